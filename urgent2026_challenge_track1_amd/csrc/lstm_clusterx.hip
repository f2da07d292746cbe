// Cluster LSTM forward with the INPUT PROJECTION FUSED: the time path of BSRNN at C2 (1,088 sequences x 401 steps per direction; espnet2 BSRNN's
// rnn_time, reference twin baseline_code/models/bsrnn_flowse.py:296-299: nn.LSTM = x W_ih^T + b_ih + h W_hh^T + b_hh).
//
// lstm_cluster.hip keeps W_hh in registers and exchanges h between the seven workgroups of a cluster; the gate pre-activations x W_ih^T + b came
// from a GEMM that wrote 2.74 GB per launch a millisecond earlier only to be read back (0.96 ms x 6 per train step).  Fusing the projection into
// THAT kernel fails on its budgets, measured in round 5: its 14 working waves have 128 registers each (16 waves per workgroup), the working path
// uses 106, and the seven k-slabs of W_ih are 28 more plus 16 for accumulators that must survive the h gather - 161, spilled (round 2 saw the same:
// 65 spills, 6.8 vs 4.6 ms); its LDS is full (156 of 160 KB).  So this kernel re-cuts the SAME cluster: SEVEN working waves per workgroup, each with
// TWO unit quads (8 hidden units x 4 gates), plus ONE helper wave - 8 waves = two per SIMD = a 256-register budget: 104 registers of W_hh + 56 of
// W_ih fragments per wave.  Per step:
//   0. x_t W_ih^T + b for the 64 rows x this wave's 32 gate columns - independent of h, issued while the first loads of the h gather are in flight;
//   1. h_{t-1} of the cluster's 64 sequences -> LDS (tag-in-data hand-off, exactly lstm_cluster.hip's), barrier 1;
//   2. + h_{t-1} W_hh^T (an A fragment read from LDS feeds both quads: half the LDS read traffic of the 14-wave form), cell update, barrier 2;
//   3. h_t -> exchange buffer (tagged); hout / saved gates / c_t leave in the NEXT step (deferred 16-byte row pieces).
// The helper wave brings x_{t+1} (64 rows x 448 bytes) by LDS-DMA into the staging tile of that step's parity - two rows per instruction at the
// bank-conflict-free pitch of 480 bytes - and stores the previous step's saved gates and c_t.  The x tile doubles as the saved-gates staging tile:
// every wave has multiplied x_t before barrier 1, the gate activations are written behind it.
// Same math / layouts / protocol as lstm_cluster.hip, and the same rounding points as the two-kernel form: x W_ih^T + b is rounded to the 16-bit
// operand format before h W_hh^T is added to it (there: the gx matrix; here: the register pairs that wait for the gather) - equal to it up to the
// summation order inside the MFMAs.  N = 196 (Np 224), H = 392 (Hp 416).
#include "urse_common.h"

namespace urse {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int XW = 7;              // working waves per workgroup
constexpr int XQ = 2;              // unit quads per working wave
constexpr int XTHR = XW * 64;      // 448 working threads
constexpr int XROWS = 64;          // sequences per cluster
constexpr int XUW = XW * XQ * 4;   // hidden units per workgroup (56)
constexpr int XNSH = 13, XNSX = 7; // k-slabs of W_hh (Hp = 416) and W_ih (Np = 224)

struct ClusterXArgs {
  const void* xn; long ldx;       // [M, ldx] 16-bit normalised input rows, K padding zero
  const void* wihq;               // [2][nq][7][64][16 B]  quad-ordered W_ih fragments (urse_lstm_pack_quads_x)
  const void* whhq;               // [2][nq][13][64][16 B] quad-ordered W_hh fragments (urse_lstm_pack_quads)
  const float* bias;              // [2][4H] f32 (dir, unit, gate): b_ih + b_hh
  void* gates; long ldg;          // out (save): bf16 gate activations
  void* hout; long ldh;
  void* hout2;                    // f16 operands: h once more in bf16 (null: not wanted)
  float* c;
  bf16_t* hx;                     // exchange [2 parity][2 dir][ncl][rows_pad][Hp]
  unsigned* err;
  int H, save;
  long inner, outer, stride;
  int n_seq, seq_len;
  int C, ncl, rows_per_cluster, rows_pad;
  unsigned g_bytes, c_bytes, h_bytes, x_bytes;
  unsigned* xws;                  // XCD-aware formation (null = static clusters): [0..7] arrivals per XCD, [8] arrivals, zeroed per launch
};

__device__ __forceinline__ void xstore_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 16);
}
__device__ __forceinline__ void xstore_plain(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 0);
}
__device__ __forceinline__ uint4 xload_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

template <typename TI, bool H2, bool SAVE>
__global__ void __launch_bounds__(XTHR + 64) lstm_fwd_clusterx_kernel(ClusterXArgs p) {
  static_assert(!H2 || __is_same(TI, f16_t), "the bf16 copy of h exists in the f16 mode only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H;
  constexpr int Hp = XNSH * 32, pitch = lds_frag_pitch(Hp * 2);          // h tile row pitch (864)
  constexpr int GP = lds_frag_pitch(XNSX * 64);                           // x / gates tile row pitch (480): conflict-free A fragment reads
  constexpr int GPC = GP / 16;                                            // 30 pieces of 16 bytes per tile row (28 carry data)
  // LDS: [h tile][h staging 0, 1][x / gates tile 0, 1][c staging 0, 1][bias][row table][flags]
  char* htile = smem;                                                     // [64][pitch]
  char* hstage0 = htile + XROWS * pitch;                                  // [2][64][XUW] TI
  char* gstage0 = hstage0 + 2 * XROWS * XUW * 2;                          // [2][64][GP]: x_t (224 channels), then the step's gate activations (56 units x 4)
  char* cstage0 = gstage0 + 2 * XROWS * GP;                               // [2][64][XUW] f32
  float* bias_s = reinterpret_cast<float*>(cstage0 + 2 * XROWS * XUW * 4);   // [XUW][4]
  int* rowtab = reinterpret_cast<int*>(bias_s + XUW * 4);                 // [64] row of (sequence, t = 0)
  int* xs = rowtab + XROWS;                                               // [12] broadcast of the cluster assignment, [12] dead flag
  unsigned* deadflag = reinterpret_cast<unsigned*>(xs + 12);

  // ---- which cluster, which member, which sequences (lstm_cluster.hip: static or XCD-aware formation; placement decides speed, never correctness)
  int dir = blockIdx.y, cl = blockIdx.x / p.C, j = blockIdx.x - cl * p.C;
  int seq0 = cl * p.rows_per_cluster, nrows_x = p.rows_per_cluster, clx = dir * p.ncl + cl;
  bool local = false;
  if (p.xws != nullptr) {
    if (tid == 0) {
      const int grid = gridDim.x * gridDim.y;
      const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;      // HW_REG_XCC_ID
      const unsigned rank = __hip_atomic_fetch_add(p.xws + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p.xws + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      bool ok = true;
      while (__hip_atomic_load(p.xws + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)grid) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { ok = false; atomicExch(p.err, 1u); break; }
      }
      int n[8], S = 0, before_full = 0, before_left = 0;
      for (int x = 0; x < 8; ++x) {
        n[x] = (int)__hip_atomic_load(p.xws + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int f = n[x] / p.C;
        if (x < (int)xcc) { before_full += f; before_left += n[x] - f * p.C; }
        S += f;
      }
      const int NC = grid / p.C, Mx = NC - S;
      const int f_me = n[xcc] / p.C;
      int mode = 0;
      if (ok && S > 0 && (S & 1) == 0 && (Mx & 1) == 0) {
        const int ns = S / 2, nm = Mx / 2;
        int rows_m = 0;
        if (nm > 0 && XROWS * ns < p.n_seq) rows_m = (p.n_seq - XROWS * ns + nm - 1) / nm;
        const int rows_s = (p.n_seq - rows_m * nm + ns - 1) / ns;
        if (rows_m <= XROWS && rows_s <= XROWS && rows_s > 0) {
          mode = 1;
          int ci, jj, lc_;
          if ((int)rank < f_me * p.C) { ci = before_full + (int)rank / p.C; jj = (int)rank % p.C; lc_ = 1; }
          else { const int li = before_left + ((int)rank - f_me * p.C); ci = S + li / p.C; jj = li % p.C; lc_ = 0; }
          const int d = ci & 1, k = lc_ ? (ci >> 1) : ((ci - S) >> 1);
          const int s0 = lc_ ? k * rows_s : ns * rows_s + k * rows_m;
          int nr = lc_ ? rows_s : rows_m;
          if (s0 + nr > p.n_seq) nr = p.n_seq - s0 > 0 ? p.n_seq - s0 : 0;
          xs[0] = 1; xs[1] = d; xs[2] = ci; xs[3] = jj; xs[4] = s0 < p.n_seq ? s0 : 0; xs[5] = nr; xs[6] = lc_;
        }
      }
      if (!mode) xs[0] = 0;
      if (!ok) xs[0] = -1;
    }
    __syncthreads();
    if (xs[0] < 0) return;                                               // the grid never assembled: flagged, nothing written
    if (xs[0] == 1) { dir = xs[1]; clx = xs[2]; j = xs[3]; seq0 = xs[4]; nrows_x = xs[5]; local = xs[6] != 0; cl = clx; }
    __syncthreads();
  }
  int seq1 = seq0 + nrows_x;
  if (seq1 > p.n_seq) seq1 = p.n_seq;
  const int nrows = seq1 > seq0 ? seq1 - seq0 : 0;
  const int nq = (H + 3) >> 2;
  const int ldg_i = (int)p.ldg, ldh_i = (int)p.ldh, ldc_i = 2 * H, stride_i = (int)p.stride, gcol_i = dir * 4 * H, hcol_i = dir * H;
  constexpr unsigned COOB = 0xFFFFF000u;

  // ---- common set-up: row table, bias of this workgroup's units, zeroed staging
  if (tid < XROWS) {
    int seq = seq0 + tid;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    rowtab[tid] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }
  for (int i = tid; i < XUW * 4; i += XTHR + 64) {
    const int u = j * XUW + (i >> 2);
    bias_s[i] = u < H ? p.bias[(long)dir * 4 * H + u * 4 + (i & 3)] : 0.f;
  }
  for (int i = tid; i < 2 * XROWS * XUW * 2 / 4; i += XTHR + 64) reinterpret_cast<unsigned*>(hstage0)[i] = 0u;      // pad units stay 0
  for (int i = tid; i < XROWS * lds_frag_pitch(XNSH * 64) / 16; i += XTHR + 64) reinterpret_cast<uint4*>(htile)[i] = make_uint4(0, 0, 0, 0);   // (K padding stays 0)
  if (tid == 0) *deadflag = 0u;
  __syncthreads();
  const int nvu = (H - j * XUW) < XUW ? (H - j * XUW > 0 ? H - j * XUW : 0) : XUW;     // valid units of this workgroup (a multiple of 8)
  const __amdgpu_buffer_rsrc_t rs_gs = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);

  if (w == XW) {
    // ================= helper wave: x_{t+1} in, saved gates / c_t of step t - 1 out; the working waves' two barriers per step =================
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NG = XROWS * GPC / 64, NC = XROWS * (XUW * 4 / 16) / 64;     // 30 gates pieces, 14 c pieces per lane and step
    constexpr int GC = XUW * 8 / 16, CC = XUW * 4 / 16;
    unsigned og[NG], oc[NC];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int idx = lane + i * 64, row = idx / GPC, cc = idx - row * GPC;
      og[i] = (row < nrows && cc < GC && cc * 2 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldg_i + (unsigned)(gcol_i + (j * XUW + cc * 2) * 4)) * 2u : COOB;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = lane + i * 64, row = idx / CC, cc = idx - row * CC;
      oc[i] = (row < nrows && cc * 4 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldc_i + (unsigned)(hcol_i + j * XUW + cc * 4)) * 4u : COOB;
    }
    // x rows: lane l holds the byte offset of row l at t = 0; one LDS-DMA instruction brings TWO rows - lanes 0 .. 27 the 28 pieces of row 2 i, lanes
    // 30 .. 57 those of row 2 i + 1 (the destination is lane-linear: lane 30 lands at byte 480 = the tile's pitch), the other lanes are masked off
    const int ldx2 = (int)p.ldx * 2;
    const unsigned xoff_row = lane < nrows ? (unsigned)rowtab[lane] * (unsigned)ldx2 : COOB;
    const bool xact = lane < 28 || (lane >= 30 && lane < 58);
    const unsigned xpiece = (unsigned)((lane < 30 ? lane : lane - 30) * 16);
    const unsigned long gb = (unsigned long)p.xn;
    typedef int rsrc4 __attribute__((ext_vector_type(4)));
    const rsrc4 rg = rsrc4{(int)(unsigned)gb, (int)(unsigned)((gb >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
    const unsigned lds_g0 = (unsigned)(size_t)gstage0;
    auto fetch = [&](int par, int toff_) {
      const unsigned soff = (unsigned)(toff_ * ldx2);
#pragma unroll
      for (int i = 0; i < XROWS / 2; ++i) {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)xoff_row, 2 * i), r1 = (unsigned)__builtin_amdgcn_readlane((int)xoff_row, 2 * i + 1);
        const unsigned vo = (lane < 30 ? r0 : r1) + xpiece;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_g0 + (unsigned)(par * (XROWS * GP) + 2 * i * GP));
#ifndef XABL_NO_DMA      // timing diagnostics (wrong results): XABL_NO_DMA, XABL_NO_HSTORE, XABL_NO_PROJ, XABL_NO_REC, XABL_NO_CELL, XABL_NO_GATHER, XABL_NO_XSTORE, XABL_NO_HOUT
        if (xact) {
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(vo), "s"(rg), "s"(soff), "s"(dst) : "memory");
        }
#else
        asm volatile("" :: "v"(vo), "s"(dst), "s"(soff));
#endif
      }
    };
    uint4 vg[NG], vc[NC];
    fetch(0, (dir ? p.seq_len - 1 : 0) * stride_i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0: x_0 is in tile 0
    // The helper's work is split over the two halves of a step so that it is never the last to arrive at a barrier: between barrier 1 and barrier 2
    // (the working waves multiply h W_hh^T and update the cells) it reads the PREVIOUS step's staged pieces into registers and issues the DMAs of
    // x_{t+1} into the tile they came from; between barrier 2 and the next barrier 1 (the working waves project x_{t+1} and gather h_t) it stores the
    // pieces.  (All of it between barrier 1 and barrier 2 - 32 DMAs + 44 LDS reads + 44 stores, ~3.3 us - made the working waves wait there.)
    bool have = false;
    int toff_st = 0;
    auto store_pieces = [&]() {
#ifndef XABL_NO_HSTORE
#pragma unroll
      for (int i = 0; i < NG; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{vg[i].x, vg[i].y, vg[i].z, vg[i].w}, rs_gs, (int)og[i], toff_st * ldg_i * 2, 0);
#pragma unroll
      for (int i = 0; i < NC; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{vc[i].x, vc[i].y, vc[i].z, vc[i].w}, rs_cs, (int)oc[i], toff_st * ldc_i * 4, 0);
#endif
    };
    auto read_pieces = [&](int pp) {
      const char* gst = gstage0 + pp * (XROWS * GP);
      const char* cst_ = cstage0 + pp * (XROWS * XUW * 4);
#pragma unroll
      for (int i = 0; i < NG; ++i) vg[i] = *reinterpret_cast<const uint4*>(gst + (lane + i * 64) * 16);
#pragma unroll
      for (int i = 0; i < NC; ++i) vc[i] = *reinterpret_cast<const uint4*>(cst_ + (lane + i * 64) * 16);
    };
    int toff_prev = 0;
    for (int step = 0; step < p.seq_len; ++step) {
      const int t = dir ? (p.seq_len - 1 - step) : step;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the stores issued behind the previous barrier 2 are done: their data registers may be rewritten
      __builtin_amdgcn_s_barrier();                                       // barrier 1 of the step
      const int pp = (step + 1) & 1;                                      // parity of the previous step = of the next one
      have = p.save && step > 0;
      if (have) read_pieces(pp);                                          // the previous step's tiles are complete behind ITS barrier 2
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the tile is in registers: x_{t+1} may overwrite it
      if (step + 1 < p.seq_len) fetch(pp, (dir ? t - 1 : t + 1) * stride_i);
      toff_st = toff_prev;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // x_{t+1} has landed: the working waves multiply it right behind barrier 2
      __builtin_amdgcn_s_barrier();                                       // barrier 2 of the step
      if (have) store_pieces();
      toff_prev = t * stride_i;
    }
    if (p.save) {                                                         // the last step's tiles
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      read_pieces((p.seq_len + 1) & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      toff_st = toff_prev;
      store_pieces();
    }
    return;
  }

  // ================= working waves =================
  // operands swapped as in lstm_cluster.hip: A = the resident weight fragment (rows = the quad's 16 gate columns), B = the h / x fragment (columns =
  // sequences): lane (lr, lc) then holds the FOUR GATES of unit lr of the quad for sequence lc of the row tile
  const int ul = lr, rl = lc;
  uint4 breg[XQ][XNSH], wreg[XQ][XNSX];
  int lu[XQ];                                                             // unit index inside the workgroup (0 .. 55); H % 56 == 0: every quad and unit is valid
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int qd = j * (XW * XQ) + w * XQ + q;
    lu[q] = (w * XQ + q) * 4 + ul;
    const char* sh = reinterpret_cast<const char*>(p.whhq) + (((long)dir * nq + qd) * XNSH) * 1024 + lane * 16;
    const char* sx = reinterpret_cast<const char*>(p.wihq) + (((long)dir * nq + qd) * XNSX) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < XNSH; ++ks) breg[q][ks] = *reinterpret_cast<const uint4*>(sh + ks * 1024);
#pragma unroll
    for (int ks = 0; ks < XNSX; ++ks) wreg[q][ks] = *reinterpret_cast<const uint4*>(sx + ks * 1024);
  }
  // c_{t-1} is read back from the previous step's c staging tile (LDS, written every step), not carried in registers: the budget is 256 and the
  // resident weights take 160
  for (int i = tid; i < 2 * XROWS * XUW; i += XTHR) reinterpret_cast<float*>(cstage0)[i] = 0.f;

  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * p.rows_pad * Hp * 2);
  const unsigned cl_bytes = (unsigned)((long)clx * p.rows_pad * Hp * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(2u * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs2 = __builtin_amdgcn_make_buffer_rsrc(H2 ? p.hout2 : p.hout, 0, (int)p.h_bytes, 0x00020000);
  const int hchunks = H / 8;                                              // 49 data chunks of 16 B per h row (H % 8 == 0)
  constexpr int HL = 7;                                                   // chunks per working thread: 64 rows x 49 = 7 x 448 (H = 392)
  static_assert(XTHR == 9 * 49 + 7, "chunk walk below: 448 = 9 rows of 49 chunks + 7");
  const int hrow0_o = tid / 49, hcc0_o = tid - hrow0_o * 49;              // chunk i of this thread: linear index tid + 448 i = (row, chunk), see hpos
  constexpr unsigned TAGM = 0x40004000u;                                  // bit 14 of both 16-bit halves: clear in |h| <= 1 (bf16 and f16)
  // the one (row, 16-byte piece) of the staged h tile this thread publishes / stores per step: 64 rows x 7 pieces = 448 = one per working thread
  constexpr int SC = XUW * 2 / 16;
  const int st_row = tid / SC, st_cc = tid - st_row * SC;
  unsigned dvo_h;
  {
    const int ucol = j * XUW + st_cc * 8;
    dvo_h = (st_row < nrows && ucol < H) ? ((unsigned)rowtab[st_row] * (unsigned)ldh_i + (unsigned)(hcol_i + ucol)) * 2u : COOB;
  }
  int toff_d = 0;
  bool have_d = false;
  auto deferred_hout = [&](int par) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const uint4 v = *reinterpret_cast<const uint4*>(hstage0 + par * (XROWS * XUW * 2) + tid * 16);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_hs, (int)dvo_h, toff_d * ldh_i * 2, 0);
    if constexpr (H2) {
      float a0, a1, a2, a3, a4, a5, a6, a7;
      unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7)},
                                             rs_hs2, (int)dvo_h, toff_d * ldh_i * 2, 0);
    }
  };
  __builtin_amdgcn_s_barrier();                                           // B0: x_0 is in tile 0 (the helper's first fetch)

  const int lr_o = lr, lc_o = lc, lu0_o = lu[0], lu1_o = lu[1];
  for (int step = 0; step < p.seq_len; ++step) {
    // The lane's index registers are made opaque once per step: every LDS / exchange address below is then recomputed from them inside the step (a
    // few dozen integer instructions) instead of being hoisted out of the time loop as ~60 loop-invariant registers - which the 256-register budget
    // (160 of it resident weights) turned into scratch spills, and a scratch reload is a vector memory operation IN FRONT of which the in-order vmcnt
    // queue waits for every gather load issued before it: the seven loads of the h gather ran as seven serial round trips.
    int lr = lr_o, lc = lc_o, hrow0 = hrow0_o, hcc0 = hcc0_o;
    int lu[XQ] = {lu0_o, lu1_o};
    asm volatile("" : "+v"(lr), "+v"(lc), "+v"(hrow0), "+v"(hcc0), "+v"(lu[0]), "+v"(lu[1]));
    const int ul = lr, rl = lc;
    (void)ul;
    auto hpos = [&](int i, int& row, int& cc) {
      const int c = hcc0 + 7 * i, wrap = c >= 49 ? 1 : 0;
      row = hrow0 + 9 * i + wrap;
      cc = c - 49 * wrap;
    };
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const int par = step & 1;
    const unsigned pprev = (unsigned)((step + 1) & 1), pcur = (unsigned)par;
    const unsigned tag_cur = (((unsigned)step >> 1) & 1u) ^ 1u;
    const unsigned tag_prev = (((unsigned)(step - 1) >> 1) & 1u) ^ 1u;
    char* xg = gstage0 + par * (XROWS * GP);
    // ---- 0. x_t W_ih^T + b (independent of h) + 1. the h gather.  The gather's first round of loads is issued in front of the projection.
    // (the projection's sums wait for the gather as 16-bit pairs of the operand format - what the two-kernel form stores in gx - : 16 registers
    //  instead of 32 across barrier 1)
    uint2 accp[4][XQ];
    auto project = [&]() {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f32x4_t accx[XQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) accx[q] = *reinterpret_cast<const f32x4_t*>(bias_s + lu[q] * 4);
        const char* xr = xg + (rt * 16 + lc) * GP + 16 * lr;
#pragma unroll
        for (int k0 = 0; k0 < XNSX; k0 += 4) {
          uint4 a[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + i < XNSX) a[i] = *reinterpret_cast<const uint4*>(xr + (k0 + i) * 64);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + i < XNSX) {
#pragma unroll
#ifndef XABL_NO_PROJ
              for (int q = 0; q < XQ; ++q) accx[q] = mfma16<TI>(wreg[q][k0 + i], a[i], accx[q]);
#else
              for (int q = 0; q < XQ; ++q) accx[q][i & 3] += __uint_as_float(wreg[q][k0 + i].x ^ a[i].x);
#endif
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) accp[rt][q] = make_uint2(pack2<TI>(accx[q][0], accx[q][1]), pack2<TI>(accx[q][2], accx[q][3]));
      }
    };
    const unsigned want = tag_prev ? TAGM : 0u;
    bool dead = *reinterpret_cast<volatile unsigned*>(deadflag) != 0u;
    {
      // the thread's chunks of the cluster's h_{t-1}: only the H / 8 data chunks of a row are walked (64 rows x 49 = 7 per working thread), all of
      // them in flight at once; the tile's K padding (chunks 49 .. 51) was zeroed once
      uint4 hn[HL];
      unsigned pend = 0u;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        hn[i] = make_uint4(0, 0, 0, 0);
#ifndef XABL_NO_GATHER
        int row, cc;
        hpos(i, row, cc);
        if (step > 0 && row < nrows && !dead) pend |= 1u << i;
#endif
      }
      auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < HL; ++i)
          if (pend & (1u << i)) {
            int row, cc;
            hpos(i, row, cc);
            hn[i] = xload_sc1(rs, pprev * plane_bytes + cl_bytes + (unsigned)(row * Hp * 2 + cc * 16));
          }
      };
      issue();
      project();                                                          // (behind the first round of loads)
      unsigned spins = 0;
      while (pend) {
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          if (pend & (1u << i)) {
            const uint4 v = hn[i];
            if ((v.x & TAGM) == want && (v.y & TAGM) == want && (v.z & TAGM) == want && (v.w & TAGM) == want) pend &= ~(1u << i);
          }
        }
        if (pend) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 20)) { atomicExch(p.err, 1u); *reinterpret_cast<volatile unsigned*>(deadflag) = 1u; pend = 0u; dead = true; }
          issue();
        }
      }
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        uint4 v = hn[i];
        v.x &= ~TAGM; v.y &= ~TAGM; v.z &= ~TAGM; v.w &= ~TAGM;
        int row, cc;
        hpos(i, row, cc);
        if (row < XROWS) *reinterpret_cast<uint4*>(htile + row * pitch + cc * 16) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // barrier 1: the h tile is whole; every wave has read x_t
#ifndef XABL_NO_HOUT
    if (have_d) deferred_hout(par ^ 1);                                   // the previous step's h rows -> hout, under this step's MFMAs
#endif
    // ---- 2. + h_{t-1} W_hh^T, cell update - as a software pipeline inside the wave: the 26 MFMAs of row tile rt + 1 are issued among the ~130 vector
    // instructions of the cell update of row tile rt (one straight-line block: every quad and unit of this geometry is valid, the save switch is a
    // template parameter; the scheduler is told the interleave).  With two waves per SIMD that leave barrier 1 together, MFMA blocks and cell updates
    // otherwise alternate in lockstep on both and the matrix pipe idles while the vector ALU is the bottleneck (ablation: 4.7 us of a 7.1 us step
    // with every memory access switched off, profiles/r05_abl_clusterx_v1.log).
    auto mm = [&](int rt, f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        float a0, a1, a2, a3;
        unpack2<TI>(accp[rt][q].x, a0, a1);
        unpack2<TI>(accp[rt][q].y, a2, a3);
        acc[q] = f32x4_t{a0, a1, a2, a3};
      }
      const char* ar = htile + (rt * 16 + lc) * pitch + 16 * lr;
#pragma unroll
      for (int ks = 0; ks < XNSH; ++ks) {
        const uint4 a = *reinterpret_cast<const uint4*>(ar + ks * 64);
#pragma unroll
#ifndef XABL_NO_REC
        for (int q = 0; q < XQ; ++q) acc[q] = mfma16<TI>(breg[q][ks], a, acc[q]);
#else
        for (int q = 0; q < XQ; ++q) acc[q][ks & 3] += __uint_as_float(breg[q][ks].x ^ a.x);
#endif
      }
    };
    auto cell = [&](int rt, const f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
      const int row = rt * 16 + rl;
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const float cprev = *reinterpret_cast<const float*>(cstage0 + (par ^ 1) * (XROWS * XUW * 4) + row * (XUW * 4) + lu[q] * 4);
#ifndef XABL_NO_CELL
        const float iv = sigmoidf_(acc[q][0]), fv = sigmoidf_(acc[q][1]), gv = tanhf_(acc[q][2]), ov = sigmoidf_(acc[q][3]);
        const float cv = fv * cprev + iv * gv;
        const float hv = ov * tanhf_(cv);
#else
        const float iv = acc[q][0], fv = acc[q][1], gv = acc[q][2], ov = acc[q][3];
        const float cv = fv * cprev + iv * gv;
        const float hv = ov * cv;
#endif
        reinterpret_cast<TI*>(hstage0 + par * (XROWS * XUW * 2))[row * XUW + lu[q]] = from_f32<TI>(hv);
        *reinterpret_cast<float*>(cstage0 + par * (XROWS * XUW * 4) + row * (XUW * 4) + lu[q] * 4) = cv;      // (also the next step's c_{t-1})
        if constexpr (SAVE) {
          uint2 gs;
          gs.x = pack2<bf16_t>(iv, fv);
          gs.y = pack2<bf16_t>(gv, ov);
          *reinterpret_cast<uint2*>(xg + row * GP + lu[q] * 8) = gs;
        }
      }
    };
    {
      f32x4_t accA[XQ], accB[XQ];
      mm(0, accA);
      mm(1, accB);
      cell(0, accA);
#ifndef XNO_SCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {      // one A fragment read + its two MFMAs per ~10 vector instructions of the cell update
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      }
#endif
      mm(2, accA);
      cell(1, accB);
#ifndef XNO_SCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 1);
      }
#endif
      mm(3, accB);
      cell(2, accA);
#ifndef XNO_SCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 2);
      }
#endif
      cell(3, accB);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // barrier 2
    // ---- 3. h_t of this workgroup's units -> exchange buffer (tagged)
    {
      const unsigned tagv = tag_cur ? TAGM : 0u;
      const int ucol = j * XUW + st_cc * 8;
      if (st_row < nrows && ucol < H && step + 1 < p.seq_len) {
        const uint4 v = *reinterpret_cast<const uint4*>(hstage0 + par * (XROWS * XUW * 2) + tid * 16);
        const uint4 vt = make_uint4(v.x | tagv, v.y | tagv, v.z | tagv, v.w | tagv);
        const unsigned xo = pcur * plane_bytes + cl_bytes + (unsigned)(st_row * Hp * 2 + ucol * 2);
#ifndef XABL_NO_XSTORE
        if (local) xstore_plain(rs, xo, vt);
        else xstore_sc1(rs, xo, vt);
#else
        asm volatile("" :: "v"(vt.x), "v"(xo));
#endif
      }
    }
    toff_d = toff;
    have_d = true;
  }
  if (have_d) deferred_hout((p.seq_len + 1) & 1);
}

// quad-ordered fragments of W_ih: block (dir, quad, slab) = 64 lanes x 16 B; lane (lr, lc): unit quad * 4 + (lc >> 2), gate lc & 3, k = slab * 32 + 8 lr + j
// (the lane map of urse_lstm_pack_quads with the input channels as k)
template <typename TI>
__device__ __forceinline__ void lstm_pack_quads_x_dev(const float* __restrict__ wih, TI* __restrict__ out, int N, int Np, int H) {
  const int nq = (H + 3) >> 2, nslab = Np / 32, G4 = 4 * H;
  const long total = (long)2 * nq * nslab * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int jj = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % nslab); r /= nslab;
    const int qd = (int)(r % nq);
    const int d = (int)(r / nq);
    const int lc = lane & 15, lr = lane >> 4;
    const int u = qd * 4 + (lc >> 2), g = lc & 3, k = ks * 32 + 8 * lr + jj;
    out[idx] = from_f32<TI>((u < H && k < N) ? wih[((long)d * G4 + g * H + u) * N + k] : 0.f);
  }
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_quads_x_kernel(const float* __restrict__ wih, TI* __restrict__ out, int N, int Np, int H) {
  lstm_pack_quads_x_dev<TI>(wih, out, N, Np, H);
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_quads_x_multi_kernel(const PackRow* __restrict__ tab, int N, int Np, int H) {
  const PackRow r = tab[blockIdx.y];
  if (r.wihq) lstm_pack_quads_x_dev<TI>(r.wih, (TI*)r.wihq, N, Np, H);
}

constexpr size_t clusterx_lds() {
  return (size_t)XROWS * lds_frag_pitch(XNSH * 64) + 2 * (size_t)XROWS * XUW * 2 + 2 * (size_t)XROWS * lds_frag_pitch(XNSX * 64) + 2 * (size_t)XROWS * XUW * 4 +
         (size_t)XUW * 16 + XROWS * sizeof(int) + 32 * sizeof(int);
}

}  // namespace urse

using namespace urse;

extern "C" int urse_lstm_pack_quads_x(const float* wih, void* out, int N, int Np, int H, int dtype, void* stream) {
  URSE_CHECK_ARG(wih && out && H > 0 && N > 0 && Np % 32 == 0 && Np >= N && (dtype == URSE_BF16 || dtype == URSE_F16), "urse_lstm_pack_quads_x: bad argument");
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_quads_x_kernel<f16_t>, dim3(256), dim3(256), 0, (hipStream_t)stream, wih, (f16_t*)out, N, Np, H);
  else hipLaunchKernelGGL(lstm_pack_quads_x_kernel<bf16_t>, dim3(256), dim3(256), 0, (hipStream_t)stream, wih, (bf16_t*)out, N, Np, H);
  URSE_CHECK_LAUNCH("urse_lstm_pack_quads_x");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_quads_x_multi(const void* table, int n_lstm, int N, int Np, int H, int dtype, void* stream) {
  URSE_CHECK_ARG(table && n_lstm > 0 && n_lstm < 65536 && H > 0 && N > 0 && Np % 32 == 0 && Np >= N && (dtype == URSE_BF16 || dtype == URSE_F16),
                 "urse_lstm_pack_quads_x_multi: bad argument");
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_quads_x_multi_kernel<f16_t>, dim3(256, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H);
  else hipLaunchKernelGGL(lstm_pack_quads_x_multi_kernel<bf16_t>, dim3(256, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H);
  URSE_CHECK_LAUNCH("urse_lstm_pack_quads_x_multi");
  return URSE_OK;
}

extern "C" int urse_lstm_clusterx_supported(int N, int Np, int H, int Hp) {
  return (N > 0 && N <= 224 && Np == 224 && Hp == 416 && H > 0 && H % XUW == 0 && H <= 416) ? 1 : 0;      // (whole workgroups of 56 units: H = 392)
}

extern "C" int urse_lstm_clusterx_fwd(const void* xn, int64_t ldx, const void* wihq, const float* bias, const void* whhq, void* gates, int64_t ldg,
                                      void* hout, int64_t ldh, float* c, void* hx, void* counters, void* err_flag, int N, int Np, int H, int Hp,
                                      int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save, int reserved_cus,
                                      int xcd_aware, int dtype, void* hout_bf16, void* stream) {
  URSE_CHECK_ARG(xn && wihq && bias && whhq && hout && hx && counters && err_flag && ((c && gates) || !save), "urse_lstm_clusterx_fwd: null pointer");
  URSE_CHECK_ARG(urse_lstm_clusterx_supported(N, Np, H, Hp), "urse_lstm_clusterx_fwd: unsupported N=%d Np=%d H=%d Hp=%d", N, Np, H, Hp);
  URSE_CHECK_ARG(dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_clusterx_fwd: operands are bf16 or f16 (dtype %d)", dtype);
  URSE_CHECK_ARG(!hout_bf16 || (dtype == URSE_F16 && ((uintptr_t)hout_bf16 % 16) == 0), "urse_lstm_clusterx_fwd: the bf16 copy of h goes with f16 operands only");
  int64_t plan[6];
  int rc = urse_lstm_cluster_plan(H, Hp, n_seq, reserved_cus, plan);       // the same clusters as urse_lstm_cluster_fwd: 7 workgroups x 56 units, 64 sequences
  if (rc) return rc;
  URSE_CHECK_ARG((!save || (ldg >= 8L * H && ldg % 8 == 0 && ((uintptr_t)gates % 16) == 0)) && ldh >= 2L * H && (ldh * 2) % 16 == 0 && ((uintptr_t)hout % 16) == 0 &&
                     ((uintptr_t)hx % 16) == 0 && ldx >= Np && (ldx * 2) % 16 == 0 && ((uintptr_t)xn % 16) == 0 && (!c || ((uintptr_t)c % 16) == 0),
                 "urse_lstm_clusterx_fwd: bad leading dimension / alignment");
  ClusterXArgs p;
  {
    const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
    const long gb = save ? rows * ldg * 2 : 0;
    URSE_CHECK_ARG(gb < 0xFFFFF000L && rows * 2L * H * 4 < 0xFFFFF000L && rows * ldh * 2 < 0xFFFFF000L && rows * ldx * 2 < 0xFFFFF000L && ldg < (1L << 31) &&
                       ldh < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                   "urse_lstm_clusterx_fwd: matrices of %ld rows exceed 32-bit byte offsets", rows);
    p.g_bytes = (unsigned)gb; p.c_bytes = (save && c) ? (unsigned)(rows * 2L * H * 4) : 0u; p.h_bytes = (unsigned)(rows * ldh * 2); p.x_bytes = (unsigned)(rows * ldx * 2);
  }
  p.xn = xn; p.ldx = ldx; p.wihq = wihq; p.whhq = whhq; p.bias = bias;
  p.gates = save ? gates : hout; p.ldg = save ? ldg : 8L * H; p.hout = hout; p.hout2 = hout_bf16; p.ldh = ldh; p.c = (save && c) ? c : reinterpret_cast<float*>(hout);
  p.hx = (bf16_t*)hx; p.err = (unsigned*)err_flag; p.H = H; p.save = save;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  p.C = (int)plan[0]; p.ncl = (int)plan[1]; p.rows_per_cluster = (int)plan[2]; p.rows_pad = (int)plan[3];
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(hx, 0, sizeof(bf16_t) * plan[4], st);            // the exchange planes start with every tag bit clear
  p.xws = nullptr;
  if (xcd_aware && plan[5] >= 9) {
    (void)hipMemsetAsync(counters, 0, sizeof(unsigned) * plan[5], st);
    p.xws = (unsigned*)counters;
  }
#define URSE_CLX_ATTR(...) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_clusterx_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  static bool once = (URSE_CLX_ATTR(bf16_t, false, true), URSE_CLX_ATTR(bf16_t, false, false), URSE_CLX_ATTR(f16_t, false, true), URSE_CLX_ATTR(f16_t, false, false),
                      URSE_CLX_ATTR(f16_t, true, true), true);
  (void)once;
#undef URSE_CLX_ATTR
  const size_t lds = clusterx_lds();
  dim3 grid(p.C * p.ncl, 2), blk(XTHR + 64);
  note_launch(URSE_KV_LSTM_FWD_CLUSTERX);
  if (dtype == URSE_F16 && hout_bf16 && save) hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<f16_t, true, true>), grid, blk, lds, st, p);
  else if (dtype == URSE_F16 && save) hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<f16_t, false, true>), grid, blk, lds, st, p);
  else if (dtype == URSE_F16) hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<f16_t, false, false>), grid, blk, lds, st, p);
  else if (save) hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<bf16_t, false, true>), grid, blk, lds, st, p);
  else hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<bf16_t, false, false>), grid, blk, lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_clusterx_fwd");
  return URSE_OK;
}
