// Generalised persistent cluster LSTM forward (bf16 | f16 operands): the protocol and math of lstm_cluster.hip's forward kernel with the
// geometry as template parameters, so that other hidden sizes fit the register file:
//   NW waves per workgroup, QPW unit quads (4 units x 4 gates = one MFMA column tile) per wave, NSLAB = Hp / 32 K slabs;
//   a wave keeps QPW x NSLAB B fragments resident and one A read from the LDS h tile feeds QPW MFMAs.
// Instantiated for the flow model (H = 768: <24, 8, 1>, 24 workgroups per cluster, 4.7 MB of W_hh per direction spread
// over their registers -- its 48..96 time-path sequences otherwise keep 3..6 streaming workgroups busy for 501 steps) and,
// as a variant, for H = 392 (<13, 8, 2>).  Hand-off: tag in data, see lstm_cluster.hip.
// TI (round 6): the operand format - the gate pre-activations read from gx, the resident W_hh fragments, the exchanged h and hout; bf16_t or f16_t
// (IEEE half: the flow DNN's forward inside north_star's 1e-3, same MFMA rate).  The tag bit (bit 14 = the exponent's top bit) is clear for
// |h| <= 1 in either format; the saved gate activations are bf16 in both.
#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int C2ROWS = 64;

struct Cluster2Args {
  void* gx; long ldg;
  const void* whhq;               // [2][nq][NSLAB][64][16 B] quad-ordered fragments (urse_lstm_pack_quads)
  void* hout; long ldh;
  float* c;
  bf16_t* hx;                     // exchange [2 parity][2 dir][ncl][64][Hp], zeroed per launch
  unsigned* err;
  int H, Hp, save;
  long inner, outer, stride;
  int n_seq, seq_len;
  int C, ncl, rows_per_cluster;
};

namespace c2 {
__device__ __forceinline__ float quad_bcast(float v, int k) {
  // value of quad-lane k, broadcast inside each group of 4 lanes (DPP quad_perm)
  int r;
  const int iv = __float_as_int(v);
  switch (k) {
    case 0: r = __builtin_amdgcn_mov_dpp(iv, 0x00, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_mov_dpp(iv, 0x55, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_mov_dpp(iv, 0xAA, 0xf, 0xf, true); break;
    default: r = __builtin_amdgcn_mov_dpp(iv, 0xFF, 0xf, 0xf, true); break;
  }
  return __int_as_float(r);
}

// write-through (sc1) 16-byte accesses to the exchange buffer: bypass this CU's L1 on loads, leave L2 on stores, so the
// hand-off needs no release / acquire fence (MI355X_MICROARCH "Valid forms": every payload store and load sc1, every
// storing wave drains vmcnt, one lane per workgroup signals with an agent-scope atomic, the poller is an sc1 load).
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 16);
}
__device__ __forceinline__ uint4 load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

}  // namespace c2

template <int NSLAB, int NW, int QPW, typename TI = bf16_t, bool SYNC3 = true>
__global__ void __launch_bounds__(NW * 64) lstm_fwd_cluster2_kernel(Cluster2Args p) {
  using namespace c2;
  constexpr int NTHR = NW * 64, UW = NW * QPW * 4, HPB = NSLAB * 64;       // units per workgroup, bytes per h row
  constexpr int CPR = HPB / 16;                                              // 16-byte chunks per h row
  constexpr int HL = (C2ROWS * CPR + NTHR - 1) / NTHR;                       // h-tile chunks per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y;
  const int cl = blockIdx.x / p.C, j = blockIdx.x - cl * p.C;
  const int H = p.H;
  constexpr int pitch = lds_frag_pitch(HPB);
  char* htile = smem;                                                        // [64][pitch]
  TI* hstage = reinterpret_cast<TI*>(smem + C2ROWS * pitch);                 // [64][UW]
  unsigned* deadflag = reinterpret_cast<unsigned*>(smem + C2ROWS * pitch + C2ROWS * UW * 2);
  const int nq = (H + 3) >> 2;
  // operands of the recurrent product swapped as in lstm_cluster.hip (A = resident W_hh fragment, B = h fragment: same register contents):
  // lane (lr, lc) gets the four gates acc[0..3] of unit lr of a quad for sequence lc of the row tile, no transpose inside the lane quads
  const int ul = lr, rl = lc;                                                // unit within quad / sequence within the row tile
  const int qd0 = (j * NW + w) * QPW;                                        // first unit quad of this wave

  uint4 breg[QPW][NSLAB];                                                    // resident B fragments
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    const int qd = qd0 + qi < nq ? qd0 + qi : 0;
    const char* src = reinterpret_cast<const char*>(p.whhq) + (((long)dir * nq + qd) * NSLAB) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) breg[qi][ks] = *reinterpret_cast<const uint4*>(src + ks * 1024);
  }
  for (int i = tid; i < C2ROWS * UW / 2; i += NTHR) reinterpret_cast<unsigned*>(hstage)[i] = 0u;   // pad units stay 0
  if (tid == 0) *deadflag = 0u;
  float cst[QPW][4];
#pragma unroll
  for (int a = 0; a < QPW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) cst[a][b] = 0.f;

  const int seq0 = cl * p.rows_per_cluster;
  int seq1 = seq0 + p.rows_per_cluster;
  if (seq1 > p.n_seq) seq1 = p.n_seq;
  const int nrows = seq1 - seq0;
  bf16_t* gx = reinterpret_cast<bf16_t*>(p.gx);
  bf16_t* hout = reinterpret_cast<bf16_t*>(p.hout);
  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * C2ROWS * HPB);
  const unsigned cl_bytes = (unsigned)(((long)dir * p.ncl + cl) * C2ROWS * HPB);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(2u * plane_bytes), 0x00020000);
  const int hchunks = H / 8;
  constexpr unsigned TAGM = 0x40004000u;

  // row of this lane for row tile rt (sequence lrow = rt*16 + rl)
  // 32-bit row indices / leading dimensions (checked on the host): an address costs one v_mad_i64_i32.  Row tiles past the cluster's
  // last row (the sampler's 48 sequences fill three of the four) are skipped altogether: nrt is wave-uniform.
  int rowb[4];
  bool rvalid[4];
  const int nrt = (nrows + 15) >> 4;
  const int ldg_i = (int)p.ldg, ldh_i = (int)p.ldh, stride_i = (int)p.stride, gcol_i = dir * 4 * H, hcol_i = dir * H, ldc_i = 2 * H;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int lrow = rt * 16 + rl;
    rvalid[rt] = lrow < nrows;
    int seq = seq0 + lrow;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    rowb[rt] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }
  auto load_gx = [&](int toff, uint2 (&dst)[QPW][4]) {
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      int u = (qd0 + qi) * 4 + ul;
      if (u >= H) u = H - 1;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        if (rt < nrt) dst[qi][rt] = *reinterpret_cast<const uint2*>(gx + ((long)(rowb[rt] + toff) * ldg_i + (gcol_i + u * 4)));
    }
  };
  uint2 gxn[QPW][4];
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) gxn[qi][rt] = make_uint2(0u, 0u);
  load_gx((dir ? p.seq_len - 1 : 0) * stride_i, gxn);
  // the pieces of the staged h tile this thread publishes / stores per step (row, 16-byte chunk), with their rows in hout
  constexpr int SC = UW * 2 / 16, SI = (C2ROWS * SC + NTHR - 1) / NTHR;
  int st_lds[SI], st_x[SI], st_grow[SI], st_u[SI];        // byte offset in hstage (-1: nothing) / in the exchange rows, row in hout, first unit
#pragma unroll
  for (int i = 0; i < SI; ++i) {
    const int idx = tid + i * NTHR;
    const int row = idx / SC, cc = idx - row * SC;
    const int ucol = j * UW + cc * 8;
    const bool ok = idx < C2ROWS * SC && row < nrows && ucol < H;
    int seq = seq0 + (row < nrows ? row : 0);
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    st_lds[i] = ok ? row * (UW * 2) + cc * 16 : -1;
    st_x[i] = row * HPB + ucol * 2;
    st_u[i] = ucol;
    st_grow[i] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }
  int h_off[HL];                                          // LDS byte offset of this thread's piece i of the h tile (-1: outside the tile)
  unsigned h_valid = 0u;                                  // pieces that carry data of a live row (polled from step 1 on)
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int idx = tid + i * NTHR;
    const int row = idx / CPR, cc = idx - row * CPR;
    h_off[i] = idx < C2ROWS * CPR ? row * pitch + cc * 16 : -1;
    if (idx < C2ROWS * CPR && row < nrows && cc < hchunks) h_valid |= 1u << i;
  }
  __syncthreads();

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const unsigned pprev = (unsigned)((step + 1) & 1), pcur = (unsigned)(step & 1);
    const unsigned tag_cur = (((unsigned)step >> 1) & 1u) ^ 1u;
    const unsigned tag_prev = (((unsigned)(step - 1) >> 1) & 1u) ^ 1u;
    // 1. h_{t-1} rows -> LDS, every 16-byte piece polled until its tags are current (piece i of this thread: chunk tid + i * NTHR of
    // the cluster's [64][HPB] exchange rows; its validity and its place in the padded LDS tile are hoisted out of the time loop)
    {
      uint4 hn[HL];
#pragma unroll
      for (int i = 0; i < HL; ++i) hn[i] = make_uint4(0, 0, 0, 0);
      unsigned pend = step > 0 ? h_valid : 0u;
      if (pend && __hip_atomic_load(deadflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) pend = 0u;
      const unsigned want = tag_prev ? TAGM : 0u;
      const unsigned xbase = pprev * plane_bytes + cl_bytes + (unsigned)tid * 16u;
      unsigned spins = 0;
      while (pend) {
#pragma unroll
        for (int i = 0; i < HL; ++i)
          if (pend & (1u << i)) hn[i] = load_sc1(rs, xbase + (unsigned)(i * NTHR * 16));
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          if (pend & (1u << i)) {
            const uint4 v = hn[i];
            if ((v.x & TAGM) == want && (v.y & TAGM) == want && (v.z & TAGM) == want && (v.w & TAGM) == want) pend &= ~(1u << i);
          }
        }
        if (pend) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 20)) { atomicExch(p.err, 1u); __hip_atomic_store(deadflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); pend = 0u; }
        }
      }
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        uint4 v = hn[i];
        v.x &= ~TAGM; v.y &= ~TAGM; v.z &= ~TAGM; v.w &= ~TAGM;
        if (h_off[i] >= 0) *reinterpret_cast<uint4*>(htile + h_off[i]) = v;
      }
    }
    uint2 gxc[QPW][4];
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) gxc[qi][rt] = gxn[qi][rt];
    __syncthreads();
    if (step + 1 < p.seq_len) load_gx((dir ? t - 1 : t + 1) * stride_i, gxn);
    // 2. gates of the wave's quads for the 64 rows
    uint2 gsave[QPW][4];
    float csave[QPW][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      if (rt >= nrt) continue;
      f32x4_t acc[QPW];
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi) acc[qi] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const char* ar = htile + (rt * 16 + lc) * pitch + 16 * lr;
#pragma unroll
      for (int ks = 0; ks < NSLAB; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(ar + ks * 64);
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi)
          acc[qi] = mfma16<TI>(breg[qi][ks], a, acc[qi]);
      }
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi) {
        const int qd = qd0 + qi;
        const bool qvalid = qd < nq;
        const bool uvalid = qvalid && qd * 4 + ul < H;
        // acc[g] = gate g of unit ul of the quad, sequence rt*16 + rl
        const float pre[4] = {acc[qi][0], acc[qi][1], acc[qi][2], acc[qi][3]};
        const uint2 gxv = gxc[qi][rt];
        float x0, x1, x2, x3;
        unpack2<TI>(gxv.x, x0, x1);
        unpack2<TI>(gxv.y, x2, x3);
        const float gi = pre[0] + x0, gf = pre[1] + x1, gg = pre[2] + x2, go = pre[3] + x3;
        const float iv = sigmoidf_(gi), fv = sigmoidf_(gf), gv = tanhf_(gg), ov = sigmoidf_(go);
        const float cv = fv * cst[qi][rt] + iv * gv;
        cst[qi][rt] = cv;
        const float hv = uvalid ? ov * tanhf_(cv) : 0.f;
        if (qvalid) hstage[(rt * 16 + rl) * UW + (w * QPW + qi) * 4 + ul] = from_f32<TI>(hv);
        gsave[qi][rt].x = (unsigned)f32_to_bf16(iv) | ((unsigned)f32_to_bf16(fv) << 16);
        gsave[qi][rt].y = (unsigned)f32_to_bf16(gv) | ((unsigned)f32_to_bf16(ov) << 16);
        csave[qi][rt] = cv;
      }
    }
    __syncthreads();
    // 3. h_t of this workgroup's units -> exchange buffer (write-through, tagged) and hout
    const unsigned tagv = tag_cur ? TAGM : 0u;
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      if (st_lds[i] < 0) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(hstage) + st_lds[i]);
      if (step + 1 < p.seq_len)
        store_sc1(rs, pcur * plane_bytes + cl_bytes + (unsigned)st_x[i], make_uint4(v.x | tagv, v.y | tagv, v.z | tagv, v.w | tagv));
      *reinterpret_cast<uint4*>(hout + ((long)(st_grow[i] + toff) * ldh_i + (hcol_i + st_u[i]))) = v;    // H % 8 == 0: whole chunks
    }
    if (p.save) {
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi) {
        const int u = (qd0 + qi) * 4 + ul;
        if (qd0 + qi < nq && u < H) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            if (rt < nrt && rvalid[rt]) {
              const int row = rowb[rt] + toff;
              *reinterpret_cast<uint2*>(gx + ((long)row * ldg_i + (gcol_i + u * 4))) = gsave[qi][rt];
              p.c[(long)row * ldc_i + (hcol_i + u)] = csave[qi][rt];
            }
        }
      }
    }
    // (round 6: the third barrier of a step is not needed for correctness - the h tile is rewritten by the next step's gather, and every reader finished its MFMAs before
    //  barrier 2; the staging tile by the next step's cell update behind ITS barrier 1, and every reader has published before it arrives there.  Measured on the flow leg,
    //  both orders twice (profiles/r06_ab_c2_sync3_v1.log): without it the train step 87.6 -> 86.4 ms, but the sampler 257.7 -> 267 ms: in the forward-only step it holds
    //  the early waves back from polling while the last ones publish.  So it stays where nothing is saved: template parameter SYNC3 = !save - as a run-time
    //  `if (!p.save)` the forward-only step lost the same 3 %, barrier and all: `r06_ab_c2_sync3_cond_v1.log`.)
    if constexpr (SYNC3) __syncthreads();
  }
}

template <int NSLAB, int NW, int QPW>
static int launch_cluster2(const Cluster2Args& p, int f16, hipStream_t st) {
  const size_t lds = (size_t)C2ROWS * lds_frag_pitch(NSLAB * 64) + (size_t)C2ROWS * NW * QPW * 4 * 2 + 16;
#define URSE_C2_GO(...) do { \
    static bool once_ = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_cluster2_kernel<NSLAB, NW, QPW, __VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); \
    (void)once_; \
    hipLaunchKernelGGL((lstm_fwd_cluster2_kernel<NSLAB, NW, QPW, __VA_ARGS__>), dim3(p.C * p.ncl, 2), dim3(NW * 64), lds, st, p); } while (0)
  if (f16 && p.save) URSE_C2_GO(f16_t, false);
  else if (f16) URSE_C2_GO(f16_t, true);
  else if (p.save) URSE_C2_GO(bf16_t, false);
  else URSE_C2_GO(bf16_t, true);
#undef URSE_C2_GO
  URSE_CHECK_LAUNCH("urse_lstm_cluster2_fwd");
  return URSE_OK;
}

static bool cluster2_geometry(int H, int Hp, int* nw, int* qpw) {
  if (Hp % 32 || H % 8 || Hp < H) return false;
  const int nslab = Hp / 32;
  if (nslab == 24) { *nw = 8; *qpw = 1; return true; }
  if (nslab == 13) { *nw = 8; *qpw = 2; return true; }
  return false;
}

}  // namespace urse

using namespace urse;

// workspace query: {C, clusters per direction, rows per cluster, hx bf16 elements}; < 0 if the shape is unsupported
extern "C" int urse_lstm_cluster2_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && H > 0 && n_seq > 0 && reserved_cus >= 0, "urse_lstm_cluster2_plan: bad argument");
  int nw, qpw;
  if (!cluster2_geometry(H, Hp, &nw, &qpw)) {
    set_error("urse_lstm_cluster2_plan: unsupported H=%d Hp=%d", H, Hp);
    return URSE_ERR_UNSUPPORTED;
  }
  const int nq = (H + 3) / 4;
  const int C = (nq + nw * qpw - 1) / (nw * qpw);
  int ncl = (device_cu_count() - reserved_cus - 4) / 2 / C;   // 2 directions * ncl * C workgroups, one per free CU with a small margin: all co-resident, or refused
  if (ncl < 1) { set_error("urse_lstm_cluster2_plan: H=%d needs %d workgroups per cluster", H, C); return URSE_ERR_UNSUPPORTED; }
  int rpc = (n_seq + ncl - 1) / ncl;
  if (rpc > C2ROWS) {
    set_error("urse_lstm_cluster2_plan: %d sequences exceed the cluster capacity", n_seq);
    return URSE_ERR_UNSUPPORTED;
  }
  ncl = (n_seq + rpc - 1) / rpc;
  plan[0] = C; plan[1] = ncl; plan[2] = rpc; plan[3] = (int64_t)2 * 2 * ncl * C2ROWS * Hp;
  return URSE_OK;
}

extern "C" int urse_lstm_cluster2_fwd(void* gx, int64_t ldg, const void* whhq, void* hout, int64_t ldh, float* c, void* hx,
                                      void* err_flag, int H, int Hp, int n_seq, int seq_len, int64_t inner, int64_t outer,
                                      int64_t stride, int save, int reserved_cus, int dtype, void* stream) {
  URSE_CHECK_ARG(gx && whhq && hout && hx && err_flag && (c || !save), "urse_lstm_cluster2_fwd: null pointer");
  URSE_CHECK_ARG(dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_cluster2_fwd: operands are bf16 or f16 (dtype %d)", dtype);
  int64_t plan[4];
  int rc = urse_lstm_cluster2_plan(H, Hp, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 4 == 0 && ldh >= 2L * H && (ldh * 2) % 16 == 0 && ((uintptr_t)hout % 16) == 0 &&
                     ((uintptr_t)hx % 16) == 0 && seq_len > 0 && inner > 0,
                 "urse_lstm_cluster2_fwd: bad leading dimension / alignment");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldh < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_cluster2_fwd: row indices must fit 32 bits");
  Cluster2Args p;
  p.gx = gx; p.ldg = ldg; p.whhq = whhq; p.hout = hout; p.ldh = ldh; p.c = c; p.hx = (bf16_t*)hx; p.err = (unsigned*)err_flag;
  p.H = H; p.Hp = Hp; p.save = save; p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  p.C = (int)plan[0]; p.ncl = (int)plan[1]; p.rows_per_cluster = (int)plan[2];
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(hx, 0, sizeof(bf16_t) * plan[3], st);      // every tag bit starts clear
  note_launch(URSE_KV_LSTM_FWD_CLUSTER2);
  if (Hp / 32 == 24) return launch_cluster2<24, 8, 1>(p, dtype == URSE_F16, st);
  return launch_cluster2<13, 8, 2>(p, dtype == URSE_F16, st);
}
