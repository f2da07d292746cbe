// "Split" LSTM backward-through-time (bf16) for few, long sequences (the time path: 1,088 sequences x 401 steps).
//
// lstm.hip's BPTT gives 16 sequences to one workgroup, which re-streams W_hh^T (1.23 MB) every step: 18 us at the
// ~70 GB/s one CU reads from its L2, on 136 of the 256 CUs.  Here NSPLIT workgroups (one per CU) share 32 sequences
// and split the REDUCTION of the recurrent product: workgroup j owns a contiguous range of hidden-unit tiles, forms
// dgates only for those units (its K range), multiplies them with its rows of W_hh (1/NSPLIT of the stream) into
// PARTIAL recurrent gradients for ALL units, keeps the partial of its own units and publishes the others; before the
// next step it adds the partials the other workgroups published for its units.
//
// Hand-off = "tag in data" (as lstm_cluster.hip): every published f32 carries the step parity of its plane in its
// mantissa LSB (the partial sums lose 1 ulp); the planes are zeroed before the launch, the first write carries 1; a
// consumer re-loads (sc1, L1-bypassing) a 16-byte chunk until its four tags are current.  No counters or fences on the
// critical path, placement-independent, bounded spins (error flag).  A plane is overwritten only by a producer that
// has consumed everybody's data of the step in between, which they published after consuming the data overwritten.
// Same math / layouts / outputs as lstm_bwd_kernel (gates: saved activations in, gate pre-activation gradients out).
#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int SW = 16;             // waves per workgroup
constexpr int SMAXO = 1;           // owned unit tiles per wave
// SMAXT (template parameter): unit tiles per wave in the product, ceil(ceil(H/16) / 16)
constexpr int STHR = SW * 64;
// sequences per cluster = 16 * SRT (template parameter: 2 for H <= 512, 1 where the LDS tiles of a bigger H need it)

struct SplitBwdArgs {
  const void* dh; long ldd;
  void* gates; long ldg;
  const float* c;
  const void* whhT;                // fragment-ordered [2][nut][nslab][64][16 B] (urse_lstm_pack)
  float* xbuf;                     // [2 planes][2 dirs][ncl][nsplit][32][UP] f32, zeroed per launch
  unsigned* err;
  int H, nsplit, ncl;
  long inner, outer, stride;
  int n_seq, seq_len;
};

__device__ __forceinline__ void split_store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 16);
}
__device__ __forceinline__ uint4 split_load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int SRT, int SMAXT>
__global__ void __launch_bounds__(STHR) lstm_bwd_split_kernel(SplitBwdArgs p) {
  constexpr int SROWS = 16 * SRT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y, ns = p.nsplit;
  const int cl = blockIdx.x / ns, j = blockIdx.x - cl * ns;
  const int H = p.H, nut = (H + 15) >> 4, G4 = 4 * H, nslab = G4 * 2 / 64, UP = nut * 16;
  // unit tiles owned by this workgroup, K slabs of their gate columns
  const int tb = nut / ns, trem = nut % ns;
  const int t0 = j * tb + min(j, trem), tcnt = tb + (j < trem ? 1 : 0), t1 = t0 + tcnt;
  const int ks0 = 2 * t0, ks1 = min(2 * t1, nslab);
  const int ownw = tcnt * 64;                            // gate columns in the LDS tile
  const int tpitch = ownw * 2 + 32;                       // 32 mod 64 bytes: the ds_read_b128 fragment reads of 16 rows spread over all banks
  char* tile = smem;                                     // [32][tpitch] dgates of the owned units (MFMA A operand)
  // row pitches of the two f32 tiles: + 4 floats, so that the four 4-row groups of a wave (lanes l / 16) fall on different banks
  // (UP and tcnt * 16 are multiples of 64 floats at H = 768: every stage write / inbuf read was a 4-way bank conflict,
  // 45 % of the kernel's LDS cycles in profiles/r02_sq_counters_v2.json)
  const int SP = UP + 4, IP = tcnt * 16 + 4;
  float* stage = reinterpret_cast<float*>(smem + SROWS * tpitch);          // [32][SP] partials to publish
  float* inbuf = stage + SROWS * SP;                     // [ns-1][32][IP] partials received
  unsigned* deadflag = reinterpret_cast<unsigned*>(inbuf + (ns - 1) * SROWS * IP);
  if (tid == 0) *deadflag = 0u;

  // this wave's tiles in the product: w, w + 8, ...; of the owned range [t0, t1) it owns those congruent to w
  const int own_first = t0 + ((w - (t0 % SW) + SW) % SW);       // first owned tile of this wave (may be >= t1)
  auto own_tile = [&](int o) -> int { return own_first + SW * o; };
  auto own_valid = [&](int o) -> bool { return own_tile(o) < t1; };

  int rowbase[SRT][4];
  const int s0 = cl * SROWS;
#pragma unroll
  for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      const bool ok = seq < p.n_seq;
      if (!ok) seq = p.n_seq - 1;
      const int rb = (int)((seq / p.inner) * p.outer + (seq % p.inner));
      rowbase[rt][r] = ok ? rb : -rb - 1;
    }
  auto rowb = [&](int rt, int r) -> int { return rowbase[rt][r] >= 0 ? rowbase[rt][r] : -(rowbase[rt][r] + 1); };

  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * nut * nslab) * 1024 + lane * 16;
  const bf16_t* dh = reinterpret_cast<const bf16_t*>(p.dh);
  bf16_t* gates = reinterpret_cast<bf16_t*>(p.gates);
  // 32-bit row indices / leading dimensions (checked on the host): an address costs one v_mad_i64_i32
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  const unsigned src_bytes = (unsigned)(SROWS * UP * 4);                      // one source workgroup's block
  const unsigned cl_bytes = (unsigned)(((long)dir * p.ncl + cl) * ns) * src_bytes;
  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * ns) * src_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, (int)(2u * plane_bytes), 0x00020000);

  float dcs[SMAXO][SRT][4], dhr[SMAXO][SRT][4], ccur[SMAXO][SRT][4];
  {
    const int toff0 = (dir ? 0 : p.seq_len - 1) * stride_i;
#pragma unroll
    for (int o = 0; o < SMAXO; ++o) {
      const int u = own_tile(o) * 16 + lc;
      const int uc = (own_valid(o) && u < H) ? u : H - 1;
#pragma unroll
      for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dcs[o][rt][r] = 0.f;
          dhr[o][rt][r] = 0.f;
          ccur[o][rt][r] = p.c[(long)(rowb(rt, r) + toff0) * ldc_i + (hcol_i + uc)];
        }
    }
  }
  // inputs of phase (B), one step ahead (they do not depend on the recurrence)
  uint2 gnx[SMAXO][SRT][4];
  float cnx[SMAXO][SRT][4];
  bf16_t dnx[SMAXO][SRT][4];
  auto load_inputs = [&](int tt) {
    const int toff_ = tt * stride_i;
    const bool first_ = dir ? (tt == p.seq_len - 1) : (tt == 0);
#pragma unroll
    for (int o = 0; o < SMAXO; ++o) {
      const int u = own_tile(o) * 16 + lc;
      if (own_valid(o) && u < H) {
#pragma unroll
        for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = rowb(rt, r) + toff_;
            gnx[o][rt][r] = *reinterpret_cast<const uint2*>(gates + ((long)row * ldg_i + (gcol_i + u * 4)));
            cnx[o][rt][r] = first_ ? 0.f : p.c[(long)(row + prev_i) * ldc_i + (hcol_i + u)];
            dnx[o][rt][r] = dh[(long)row * ldd_i + (hcol_i + u)];
          }
      }
    }
  };
  load_inputs(dir ? 0 : p.seq_len - 1);
  __syncthreads();

  const int in_cpr = tcnt * 4;                            // 16-byte chunks per received row (tcnt*16 f32)
  const int in_chunks = (ns - 1) * SROWS * in_cpr;
  // the (at most four) received chunks of this thread: offset in the exchange planes and in `inbuf`, once (the run-time divisions
  // by the split geometry cost more per step than the cell update)
  constexpr int NCH = 4;                                  // chunks per thread (in_chunks <= 4 * 1024 for H <= 512 / 768 with 16 rows)
  unsigned in_off[NCH];
  int in_dst[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int idx = tid + i * STHR;
    in_off[i] = 0u;
    in_dst[i] = -1;
    if (idx < in_chunks) {
      const int s_ = idx / (SROWS * in_cpr), rem = idx - s_ * (SROWS * in_cpr);
      const int row = rem / in_cpr, cc = rem - row * in_cpr;
      const int js = s_ < j ? s_ : s_ + 1;
      in_off[i] = cl_bytes + (unsigned)js * src_bytes + (unsigned)((row * UP + t0 * 16 + cc * 4) * 4);
      in_dst[i] = (s_ * SROWS + row) * IP + cc * 4;
    }
  }
  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? step : (p.seq_len - 1 - step);
    const int toff = t * stride_i;
    const unsigned pprev = (unsigned)((step + 1) & 1), pcur = (unsigned)(step & 1);
    const unsigned tag_cur = (((unsigned)step >> 1) & 1u) ^ 1u;
    const unsigned tag_prev = (((unsigned)(step - 1) >> 1) & 1u) ^ 1u;

    // (A) partial recurrent gradients of my units published by the other workgroups during the previous step
    if (step > 0) {
      uint4 v[NCH];
      unsigned pend = 0u;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        v[i] = make_uint4(0, 0, 0, 0);
        if (in_dst[i] >= 0) pend |= 1u << i;
      }
      const unsigned pbase = pprev * plane_bytes;
      if (pend && *reinterpret_cast<volatile unsigned*>(deadflag)) pend = 0u;
#ifdef SABL_NO_WAIT
      pend = 0u;
#endif
      unsigned spins = 0;
      while (pend) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if (pend & (1u << i)) v[i] = split_load_sc1(rs, pbase + in_off[i]);
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if ((pend & (1u << i)) && ((v[i].x & 1u) == tag_prev) && ((v[i].y & 1u) == tag_prev) && ((v[i].z & 1u) == tag_prev) &&
              ((v[i].w & 1u) == tag_prev))
            pend &= ~(1u << i);
        if (pend) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 20)) { atomicExch(p.err, 1u); *reinterpret_cast<volatile unsigned*>(deadflag) = 1u; pend = 0u; }
        }
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (in_dst[i] >= 0) *reinterpret_cast<uint4*>(inbuf + in_dst[i]) = v[i];
    }
    __syncthreads();

    // (B) dgates of the owned units
#pragma unroll
    for (int o = 0; o < SMAXO; ++o) {
      const int ut = own_tile(o);
      if (!own_valid(o)) continue;
      const int u = ut * 16 + lc;
      const int tcol = ((ut - t0) * 16 + lc) * 8;          // byte column of this unit's 4 gates in the LDS tile
      if (u < H) {
#pragma unroll
        for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lrow = rt * 16 + lr * 4 + r;
            const uint2 gv2 = gnx[o][rt][r];
            const float iv = __uint_as_float(gv2.x << 16), fv = __uint_as_float(gv2.x & 0xffff0000u);
            const float gv = __uint_as_float(gv2.y << 16), ov = __uint_as_float(gv2.y & 0xffff0000u);
            float rec = dhr[o][rt][r];
            if (step > 0) {
              for (int s = 0; s < ns - 1; ++s) rec += inbuf[(s * SROWS + lrow) * IP + (ut - t0) * 16 + lc];
            }
            const float dht = bf16_to_f32(dnx[o][rt][r]) + rec;
            const float tc = tanhf_(ccur[o][rt][r]);
            const float dct = dcs[o][rt][r] + dht * ov * (1.f - tc * tc);
            const float d0 = dct * gv * iv * (1.f - iv);
            const float d1 = dct * cnx[o][rt][r] * fv * (1.f - fv);
            const float d2 = dct * iv * (1.f - gv * gv);
            const float d3 = dht * tc * ov * (1.f - ov);
            dcs[o][rt][r] = dct * fv;
            ccur[o][rt][r] = cnx[o][rt][r];           // c_{t-1} is the next processed step's c_t
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
            pk.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
            *reinterpret_cast<uint2*>(tile + lrow * tpitch + tcol) = pk;
#ifndef SABL_NO_ST
            if (rowbase[rt][r] >= 0) *reinterpret_cast<uint2*>(gates + ((long)(rowbase[rt][r] + toff) * ldg_i + (gcol_i + u * 4))) = pk;
#endif
          }
      } else {
        // pad units of the last tile: their gate columns must read as zero in the product
#pragma unroll
        for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<uint2*>(tile + (rt * 16 + lr * 4 + r) * tpitch + tcol) = make_uint2(0u, 0u);
      }
    }
    if (step + 1 == p.seq_len) break;
    load_inputs(dir ? t + 1 : t - 1);         // next step's (B) inputs: in flight during (C), (D) and the hand-off wait
    __syncthreads();

    // (C) partial dh_{t-1}[rows, all units] = dgates[rows, my K range] * W_hh[my K range, units]
#pragma unroll 1
    for (int ui = 0; ui < SMAXT; ++ui) {
      const int ut = w + SW * ui;
      if (ut < nut) {
        f32x4_t acc[SRT];
#pragma unroll
        for (int rt = 0; rt < SRT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* wr = whhT + ((long)ut * nslab) * 1024;
        const char* ar = tile + lc * tpitch + 16 * lr;
        constexpr int KB = 9;
        for (int k0 = ks0; k0 < ks1; k0 += KB) {
          uint4 b[KB];
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (k0 + i < ks1) ? k0 + i : ks1 - 1;
#ifdef SABL_NO_W
            b[i] = make_uint4(ks, ks, ks, ks);
#else
            b[i] = *reinterpret_cast<const uint4*>(wr + ks * 1024);
#endif
          }
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            if (k0 + i < ks1) {
#pragma unroll
              for (int rt = 0; rt < SRT; ++rt) {
                const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * tpitch + (k0 + i - ks0) * 64);
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                  __builtin_bit_cast(bf16x8_t, b[i]), acc[rt], 0, 0, 0);
              }
            }
          }
        }
        const bool mine = ut >= t0 && ut < t1;
        if (mine) {
          const int o = (ut - own_first) / SW;
#pragma unroll
          for (int oo = 0; oo < SMAXO; ++oo)
            if (oo == o) {
#pragma unroll
              for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dhr[oo][rt][r] = acc[rt][r];
            }
        } else {
#pragma unroll
          for (int rt = 0; rt < SRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(rt * 16 + lr * 4 + r) * SP + ut * 16 + lc] = acc[rt][r];
        }
      }
    }
    __syncthreads();

    // (D) publish the partials of the units the other workgroups own: tagged, write-through, 16 bytes per lane
    {
      const unsigned tagv = tag_cur;
      const int cpr = UP / 4;                              // 16-byte chunks per row; wave w publishes rows w, w + 16, ...
      const unsigned pub = pcur * plane_bytes + cl_bytes + (unsigned)j * src_bytes;
      for (int row = w; row < SROWS; row += SW) {
        for (int cc = lane; cc < cpr; cc += 64) {
          const int ut = cc >> 2;
          if (ut >= t0 && ut < t1) continue;
          uint4 v = *reinterpret_cast<const uint4*>(stage + row * SP + cc * 4);
          v.x = (v.x & ~1u) | tagv; v.y = (v.y & ~1u) | tagv; v.z = (v.z & ~1u) | tagv; v.w = (v.w & ~1u) | tagv;
#ifndef SABL_NO_PUB
          split_store_sc1(rs, pub + (unsigned)((row * UP + cc * 4) * 4), v);
#endif
        }
      }
    }
    // the next step's (A) writes inbuf (last read in (B) above, two barriers ago) and (B) rewrites the tile (last read
    // in (C), one barrier ago); `stage` is rewritten in (C) of the next step, two barriers after these reads
  }
}

}  // namespace urse

using namespace urse;

static size_t split_lds(int nut, int ns, int rows) {
  const int tmax = (nut + ns - 1) / ns;
  return (size_t)rows * (tmax * 128 + 32) + (size_t)rows * (nut * 16 + 4) * 4 + (size_t)(ns - 1) * rows * (tmax * 16 + 4) * 4 + 16;
}

// workspace query: {nsplit, clusters per direction, xbuf f32 elements, rows per cluster}; < 0 if the shape has no split kernel
extern "C" int urse_lstm_split_plan(int H, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && H > 0 && n_seq > 0 && reserved_cus >= 0, "urse_lstm_split_plan: bad argument");
  const int nut = (H + 15) / 16;
  if (H % 8 != 0 || nut > 3 * SW) {
    set_error("urse_lstm_split_plan: unsupported H=%d", H);
    return URSE_ERR_UNSUPPORTED;
  }
  // prefer 32-row clusters; fall back to 16 rows where the LDS tiles of a large H do not fit; as many splits (<= 3 / 12) as keep
  // every workgroup co-resident, give each at least two unit tiles and at most one owned tile per wave
  for (int rows = 32; rows >= 16; rows -= 16) {
    const int ncl = (n_seq + rows - 1) / rows;
    // 16-row clusters: up to 12 splits (flow model, H = 768, B = 2: train step 103.3 ms with 6, 97.3 with 8, 89.4 with 12, 89.2 with 16 -
    // the exchange per workgroup stays ~45 KB per step whatever the split, the weight pass shrinks with it)
    static const int max16 = getenv("URSE_SPLIT_MAX_NS") ? atoi(getenv("URSE_SPLIT_MAX_NS")) : 12;
    for (int ns = rows == 32 ? 3 : max16; ns >= 2; --ns) {
      const int tmax = (nut + ns - 1) / ns;
      if (2L * ncl * ns > device_cu_count() - reserved_cus - 6 || nut < 2 * ns || tmax > SMAXO * SW) continue;   // all workgroups resident
      if (split_lds(nut, ns, rows) > 160 * 1024) continue;
      plan[0] = ns; plan[1] = ncl; plan[2] = (int64_t)2 * 2 * ncl * ns * rows * nut * 16; plan[3] = rows;
      return URSE_OK;
    }
  }
  set_error("urse_lstm_split_plan: unsupported H=%d n_seq=%d (%d CUs reserved)", H, n_seq, reserved_cus);
  return URSE_ERR_UNSUPPORTED;
}

template <int SRT, int SMAXT>
static int launch_split(const SplitBwdArgs& p, size_t lds, hipStream_t st) {
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_split_kernel<SRT, SMAXT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  hipLaunchKernelGGL((lstm_bwd_split_kernel<SRT, SMAXT>), dim3(p.ncl * p.nsplit, 2), dim3(STHR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_split_bwd");
  return URSE_OK;
}

extern "C" int urse_lstm_split_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT,
                                   void* xbuf, void* err_flag, int H, int n_seq, int seq_len, int64_t inner, int64_t outer,
                                   int64_t stride, int reserved_cus, void* stream) {
  URSE_CHECK_ARG(dh && gates && c && whhT && xbuf && err_flag, "urse_lstm_split_bwd: null pointer");
  int64_t plan[4];
  int rc = urse_lstm_split_plan(H, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 4 == 0 && ldd >= 2L * H && ((uintptr_t)xbuf % 16) == 0 && seq_len > 0 && inner > 0,
                 "urse_lstm_split_bwd: bad leading dimension / alignment");
  URSE_CHECK_ARG(plan[2] * 4 < (1L << 31), "urse_lstm_split_bwd: exchange buffer exceeds the 2 GiB buffer-descriptor range");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldd < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_split_bwd: row indices must fit 32 bits");
  SplitBwdArgs p;
  p.dh = dh; p.ldd = ldd; p.gates = gates; p.ldg = ldg; p.c = c; p.whhT = whhT; p.xbuf = (float*)xbuf;
  p.err = (unsigned*)err_flag; p.H = H; p.nsplit = (int)plan[0]; p.ncl = (int)plan[1];
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(xbuf, 0, sizeof(float) * plan[2], st);   // every tag bit starts clear (see the hand-off protocol)
  const int nut = (H + 15) / 16, rows = (int)plan[3];
  const size_t lds = split_lds(nut, p.nsplit, rows);
  const int maxt = (nut + SW - 1) / SW;
  note_launch(URSE_KV_LSTM_BWD_SPLIT);
  if (rows == 32) {
    if (maxt <= 2) return launch_split<2, 2>(p, lds, st);
    return launch_split<2, 3>(p, lds, st);
  }
  if (maxt <= 2) return launch_split<1, 2>(p, lds, st);
  return launch_split<1, 3>(p, lds, st);
}
