// Cluster LSTM forward with the INPUT PROJECTION FUSED: the time path of BSRNN at C2 (1,088 sequences x 401 steps per direction; espnet2 BSRNN's
// rnn_time, reference twin baseline_code/models/bsrnn_flowse.py:296-299: nn.LSTM = x W_ih^T + b_ih + h W_hh^T + b_hh).
//
// lstm_cluster.hip keeps W_hh in registers and exchanges h between the seven workgroups of a cluster; the gate pre-activations x W_ih^T + b came
// from a GEMM that wrote 2.74 GB per launch a millisecond earlier only to be read back (0.96 ms x 6 per train step).  Fusing the projection into
// THAT kernel fails on its budgets, measured in round 5: its 14 working waves have 128 registers each (16 waves per workgroup), the working path
// uses 106, and the seven k-slabs of W_ih are 28 more plus 16 for accumulators that must survive the h gather - 161, spilled (round 2 saw the same:
// 65 spills, 6.8 vs 4.6 ms); its LDS is full (156 of 160 KB).  So this kernel re-cuts the SAME cluster: SEVEN working waves per workgroup, each with
// TWO unit quads (8 hidden units x 4 gates), plus ONE helper wave - 8 waves = two per SIMD = a 256-register budget: 104 registers of W_hh + 48 of
// W_ih fragments per wave.  Per step:
//   0. x_t W_ih^T + b for the 64 rows x this wave's 32 gate columns - independent of h, issued while the loads of the h gather are in flight (six of
//      the seven k-slabs of W_ih: the seventh - input channels 192 .. 195 + padding - sits in the K padding of W_hh's last slab and the h tile's
//      chunk 49 carries x_t[192 .. 199], so it costs no register, no MFMA and no LDS read of its own);
//   1. h_{t-1} of the cluster's 64 sequences -> LDS (tag-in-data hand-off, lstm_cluster.hip's protocol), barrier 1;
//   2. + h_{t-1} W_hh^T (an A fragment read from LDS feeds both quads), cell update, barrier 2;
//   3. h_t -> exchange buffer (tagged); hout leaves in the NEXT step (deferred 16-byte row pieces).
// Every wave brings four of the step's 32 LDS-DMA instructions of x_{t+1} (two 448-byte rows each, at the bank-conflict-free pitch of 480 bytes)
// right behind barrier 1; the helper wave copies x[192 .. 199] into the h tile and stores the step's saved gates and c_t behind barrier 2.  The x
// tile doubles as the saved-gates staging tile: every wave has multiplied x_t before barrier 1, the gate activations are written behind it.
// What the in-kernel cycle stamps (XSTAMP, scripts/abl_clusterx.py) and the ablations decided, in the order found:
//   * addresses: every LDS / exchange offset is the lane id (made opaque once per step and per phase, against loop-invariant hoisting -> spills)
//     times a constant + an instruction's immediate; the exchange planes keep rows at the LDS tile's pitch and the chunk walk is "nine rows of 49
//     chunks per pass" so that a pass is an immediate too: 730 -> 470 vector instructions per wave and step;
//   * the hand-off waits per WAVE, not per lane (one ballot per round), and looks at the gather once in the middle of the projection;
//   * A fragments are read three (h tile) / four (x tile) ahead of their MFMAs: written as read -> MFMA -> MFMA the compiler kept that order and
//     every k-slab waited out an LDS round trip;
//   * the helper wave alone issuing the 32 DMAs took 9,000 of a step's 15,700 cycles (an LDS-DMA instruction costs its wave 100 - 300 cycles) and
//     the working waves waited 3,400 cycles at barrier 2 for x_{t+1}: the DMAs are spread over all eight waves;
//   * a 16-byte buffer store reads its data registers some time after it has issued: an LDS read a few instructions behind it into the same
//     registers can land first (first dword of the stored piece replaced, in the waves that issue last) - the registers stay occupied until the
//     MFMA block behind the store has issued (deferred_hout).
// Round 6: (i) ROUNDS - more sequences than the co-resident clusters hold (the band path: rnn_band of the same reference lines, 12,832 sequences x 34 steps per
// direction) go through the resident clusters 64 per cluster and round (urse_lstm_clusterx_plan; the protocol's step counter runs on across rounds, a round's first
// step waits for the previous round's last publication and starts from h = 0): 2.46 against the row-wave kernel's 2.90 ms per launch in the step, 0.46 against 1.23 ms
// at B = 4; (ii) phase 2 of a step (barrier 1 -> barrier 2) interleaved by hand, one row tile ahead (XPIPE), and the projection's fragment reads as one ring
// (XPROJ_RING); no spilled register.  Per-wave stamps (profiles/r06_abl_clusterx_waves_v1.log): the second-dispatched wave of each SIMD pair is the last at barrier 2
// (6,670 cycles against 5,100 / 4,600), and the x rows' DMA chain (issue 2,300 + landing 3,500) is right behind it - phase 2 without ANY arithmetic is 0.3 us shorter.
// Same math / layouts / protocol as lstm_cluster.hip.  Round 5 kept the two-kernel form's rounding point (x W_ih^T + b rounded to the 16-bit operand format before h W_hh^T is
// added: there the gx matrix, here register pairs that waited for the gather; four channels unrounded); round 6 keeps the sum in f32 (XPROJ_F32) - nn.LSTM's own accumulation.
// N = 196 (Np 224), H = 392 (Hp 416).
#include "urse_common.h"

namespace urse {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int XW = 7;              // working waves per workgroup
constexpr int XQ = 2;              // unit quads per working wave
constexpr int XTHR = XW * 64;      // 448 working threads
constexpr int XROWS = 64;          // sequences per cluster
constexpr int XUW = XW * XQ * 4;   // hidden units per workgroup (56)
constexpr int XNSH = 13, XNSX = 7; // k-slabs of W_hh (Hp = 416) and W_ih (Np = 224)
constexpr int XNSP = 6;            // k-slabs of W_ih the PROJECTION multiplies: the seventh (input channels 192 .. 195 + padding) rides in the free k positions
                                   // 392 .. 399 of the recurrent product's last slab (eight resident registers and eight MFMAs per wave and step less)
#ifndef XFETCH_POS
#define XFETCH_POS 0
#endif
#ifndef XHSLEEP
#define XHSLEEP 48                 // s_sleep units (64 cycles) the helper waits behind barrier 2 before it reads and stores the step's gates and c_t: the publication of h_t and the gather go first (in the step: 14.9 ms of kernel per train step with 24, 14.3 with 48, 14.4 with 64, 14.9 with 80, 16.1 with 100; profiles/r05_ab_clusterx_params_v2.log)
#endif
#ifndef XAD
#define XAD 3                      // A fragments of the h tile in flight ahead of their MFMAs
#endif
#ifndef XDW
#define XDW 2                      // LDS-DMA instructions of x_{t+1} per working wave and step, the helper takes the other 18 (in the step: 14.70 ms of cluster forward per train step with 2, 14.84 with 3, 15.16 with 4, profiles/r05_ab_xdw_v1.log)
#endif
#ifndef XDWA
#define XDWA XDW                   // ... of waves 0 .. 3
#endif
#ifndef XDWB
#define XDWB XDW                   // ... of waves 4 .. 6
#endif
#ifndef XMIDPOLL
#define XMIDPOLL 0x2               // behind which row tiles of the projection the wave looks at the gather (bit rt); round 6 re-measured with the shorter projection: one look behind tile 0, 1 or 2 is the same step (28.1 - 28.6 ms of forward per train step), two or three looks cost 10 ms (profiles/r06_ab_midpoll_v1.log)
#endif
#ifndef XPROJ_F32
#define XPROJ_F32 1                 // x W_ih^T + b waits for the gather in f32 (round 6: the hand-interleaved phase 2 left the 16 registers; no pack / unpack: 48 vector instructions per wave
                                    // and step less on a SIMD whose vector ISSUE is phase 2's bound - forward 28.4 -> 27.7 ms per train step, four A/B pairs).  The sum is then not rounded
                                    // to the 16-bit operand format before h W_hh^T is added: what nn.LSTM's f32 accumulation does, one rounding less than the two-kernel form (whose gx
                                    // matrix is stored in 16 bits): h differs from it by 1.1e-4 on average (bf16), 1.4e-5 (f16).  0: the packed pairs of round 5
#endif
#ifndef XPROJ_RING
#define XPROJ_RING 1                // the projection's fragment reads as one ring across the row tiles (round 6: top -> projected 2,620 -> 2,370 cycles, forward 28.2 -> 27.5 ms per train step; with a scheduling fence per k-slab and six ahead: 2,200 cycles, 28.4 ms), 0: groups of XPD per row tile
#endif
#ifndef XPIPE
#define XPIPE 1                    // phase 2 of a step interleaved by hand (round 6), 0: the compiler's order
#endif
#ifndef XPD
#define XPD 4                      // A fragments of the x tile read ahead in the projection (registers: 4 each)
#endif

struct ClusterXArgs {
  const void* xn; long ldx;       // [M, ldx] 16-bit normalised input rows, K padding zero
  const void* wihq;               // [2][nq][7][64][16 B]  quad-ordered W_ih fragments (urse_lstm_pack_quads_x)
  const void* whhq;               // [2][nq][13][64][16 B] quad-ordered W_hh fragments (urse_lstm_pack_quads)
  const float* bias;              // [2][4H] f32 (dir, unit, gate): b_ih + b_hh
  void* gates; long ldg;          // out (save): bf16 gate activations
  void* hout; long ldh;
  void* hout2;                    // f16 operands: h once more in bf16 (null: not wanted)
  float* c;
  bf16_t* hx;                     // exchange [2 parity][2 dir][ncl][rows_pad][432 = the LDS tile's pitch]
  unsigned* err;
  int H, save;
  long inner, outer, stride;
  int n_seq, seq_len;
  int C, ncl, rows_per_cluster, rows_pad;
  int rounds;                     // > 1: more sequences than the co-resident clusters hold at once - every cluster takes 64 per round, round after round
  unsigned g_bytes, c_bytes, h_bytes, x_bytes;
  unsigned* xws;                  // XCD-aware formation (null = static clusters): [0..7] arrivals per XCD, [8] arrivals, zeroed per launch
};

#if XPROJ_F32
#define PROJ_KEEP(v) (v)
#define PROJ_TAKE(v) (v)
#else
#define PROJ_KEEP(v) make_uint2(pack2<TI>((v)[0], (v)[1]), pack2<TI>((v)[2], (v)[3]))
#define PROJ_TAKE(v) proj_take<TI>(v)
template <typename TI> __device__ __forceinline__ f32x4_t proj_take(uint2 v) {
  float a0, a1, a2, a3;
  unpack2<TI>(v.x, a0, a1);
  unpack2<TI>(v.y, a2, a3);
  return f32x4_t{a0, a1, a2, a3};
}
#endif

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

#ifdef XSTAMP      // timing diagnostics: cycle stamps of workgroup (XSTAMP_WG, 0): slots 0 .. 7 working wave 0, 8 .. 11 the helper wave, [step][16]
__device__ unsigned long long g_xstamps[512 * 16];
#define XST(slot) do { if (stamp_on && step < 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_xstamps[step * 16 + (slot)] = t_; } } while (0)
#else
#define XST(slot) do { } while (0)
#endif

// NT (round 6): row tiles of 16 sequences a cluster multiplies per step - 4 (64 sequences per cluster), or 1 / 2 / 3 for launches whose plan gives a cluster at most 16 / 32 / 48
// (small batches: the time path of up to 7 / 14 / 21 utterances); everything per row - gather passes, x DMAs, projection, recurrent product, cell update, the helper's
// pieces - is bounded by it at compile time, a sequence's arithmetic is the same in every instance (bit-identical results)
template <typename TI, bool H2, bool SAVE, int NT = 4>
__global__ void __launch_bounds__(XTHR + 64) lstm_fwd_clusterx_kernel(ClusterXArgs p) {
  static_assert(NT >= 1 && NT <= 4, "row tiles per step");
  static_assert(!H2 || __is_same(TI, f16_t), "the bf16 copy of h exists in the f16 mode only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H;
#ifdef XSTAMP
#ifndef XSTAMP_W
#define XSTAMP_W 0                 // which working wave is stamped
#endif
  const bool stamp_on = blockIdx.x == XSTAMP && blockIdx.y == 0 && (w == XSTAMP_W || w == XW);
#endif
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
  unsigned* xrow = reinterpret_cast<unsigned*>(xs + 32);                  // [64] byte offset of the x row of (sequence, t = 0), out of range for a dead row
  unsigned* hrow = xrow + XROWS;                                          // [448] byte offset of the working thread's h piece in hout at t = 0

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
        int rows_s = (p.n_seq - rows_m * nm + ns - 1) / ns;
        if (p.rounds > 1) rows_s = rows_m = XROWS;                         // (rounds: 64 sequences per cluster and round, the single-XCD clusters first)
        if (rows_m <= NT * 16 && rows_s <= NT * 16 && rows_s > 0) {         // (what this instance's row tiles hold; otherwise the static clusters of the plan)
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
  // ROUNDS (the band path: 12,832 sequences per direction, 34 steps): the cluster keeps its weights and takes 64 sequences per round, ncl * 64 sequences
  // apart.  The step counter of the hand-off runs on across rounds (gs): every step publishes, also a round's last one, and the first step of the next
  // round WAITS for that publication before it starts from h = 0 - a member that ran ahead would otherwise overwrite the plane its partner still gathers.
  const int seq0_first = seq0, rstride = p.ncl * XROWS;
  const int nrounds = (p.rounds > 1 && seq0_first < p.n_seq) ? (p.n_seq - seq0_first + rstride - 1) / rstride : 1;
  const int nq = (H + 3) >> 2;
  const int ldg_i = (int)p.ldg, ldh_i = (int)p.ldh, ldc_i = 2 * H, stride_i = (int)p.stride, gcol_i = dir * 4 * H, hcol_i = dir * H;
  constexpr unsigned COOB = 0xFFFFF000u;

  // ---- common set-up: bias of this workgroup's units, zeroed staging
  for (int i = tid; i < XUW * 4; i += XTHR + 64) {
    const int u = j * XUW + (i >> 2);
    bias_s[i] = u < H ? p.bias[(long)dir * 4 * H + u * 4 + (i & 3)] : 0.f;
  }
  for (int i = tid; i < 2 * XROWS * XUW * 2 / 4; i += XTHR + 64) reinterpret_cast<unsigned*>(hstage0)[i] = 0u;      // pad units stay 0
  for (int i = tid; i < XROWS * lds_frag_pitch(XNSH * 64) / 16; i += XTHR + 64) reinterpret_cast<uint4*>(htile)[i] = make_uint4(0, 0, 0, 0);   // (K padding stays 0)
  if (tid == 0) *deadflag = 0u;
  const int nvu = (H - j * XUW) < XUW ? (H - j * XUW > 0 ? H - j * XUW : 0) : XUW;     // valid units of this workgroup (a multiple of 8)
  const __amdgpu_buffer_rsrc_t rs_gs = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  // ---- x rows by LDS-DMA: one instruction brings TWO rows - lanes 0 .. 27 the 28 pieces of row 2 i, lanes 30 .. 57 those of row 2 i + 1 (the destination
  // is lane-linear: lane 30 lands at byte 480 = the tile's pitch), the other lanes are masked off.  EVERY wave issues some of the 32 instructions of a
  // step (XDW per working wave, the rest the helper) right behind barrier 1: issued by the helper wave alone they took 9,000 cycles of a 15,700-cycle step (in-kernel stamps,
  // profiles/r05_abl_clusterx_v9_stamps.log: an LDS-DMA instruction costs its wave 100 - 300 cycles of issue) and the working waves waited 3,400 cycles
  // at barrier 2 for x_{t+1} to land.  The byte offsets of the 64 rows at t = 0 sit in an LDS table.
  const int ldx2 = (int)p.ldx * 2;
  typedef int rsrc4 __attribute__((ext_vector_type(4)));
  const unsigned long gbx = (unsigned long)p.xn;
  const rsrc4 rg = rsrc4{(int)(unsigned)gbx, (int)(unsigned)((gbx >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
  const unsigned lds_g0 = (unsigned)(size_t)gstage0;
  auto fetch4 = [&](int par, int toff_, int lane_) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)(toff_ * ldx2);
    const bool xact = lane_ < 28 || (lane_ >= 30 && lane_ < 58);
    const unsigned xpiece = (unsigned)((lane_ < 30 ? lane_ : lane_ - 30) * 16);
    // (XDW instructions per working wave, the other 32 - 7 XDW by the helper, which has the time: it reaches barrier 2 ~1,700 cycles before the working
    //  waves when all eight take four)
    // round 6: waves 0 .. 3 take XDWA each, waves 4 .. 6 XDWB (the second-dispatched half is the arbitration loser of every SIMD pair and the LAST at
    // barrier 2 by ~1,500 cycles - per-wave stamps; an LDS-DMA costs its wave 100 - 130 cycles of issue), the helper the rest
    constexpr int HDW = XROWS / 2 - 4 * XDWA - (XW - 4) * XDWB;
    static_assert(HDW >= 0, "DMA split");
    constexpr int MAXDW = XDWA > XDWB ? (XDWA > HDW ? XDWA : HDW) : (XDWB > HDW ? XDWB : HDW);
    const int ndw = w == XW ? HDW : (w < 4 ? XDWA : XDWB);
    const int pr0 = w == XW ? 4 * XDWA + (XW - 4) * XDWB : (w < 4 ? w * XDWA : 4 * XDWA + (w - 4) * XDWB);
#pragma unroll
    for (int i = 0; i < MAXDW; ++i) {
      if (i >= ndw) continue;
      const int pr = pr0 + i;                                              // row pair
      if (NT < 4 && pr >= NT * 8) continue;                                // (rows past the instance's row tiles are never read)
      const unsigned vo = xrow[2 * pr + (lane_ < 30 ? 0 : 1)] + xpiece;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_g0 + (unsigned)(par * (XROWS * GP) + 2 * pr * GP));
#ifndef XABL_NO_DMA      // timing diagnostics (wrong results): XABL_NO_DMA, XABL_NO_HSTORE, XABL_NO_PROJ, XABL_NO_REC, XABL_NO_AREAD (no A fragment reads), XABL_NO_CELL, XABL_NO_GATHER, XABL_NO_XSTORE, XABL_NO_HOUT
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
  if (w == XW) {
    // ================= helper wave: saved gates / c_t of every step out; the working waves' two barriers per step =================
    // (its own copy of the round loop, in front of the working waves' weight loads: inside ONE loop the compiler keeps the 160 weight registers live across the
    //  helper's 220 registers of store offsets and pieces - 438 scratch instructions)
    for (int rnd = 0; rnd < nrounds; ++rnd) {
    const int seq0r = seq0_first + rnd * rstride;
    int seq1 = seq0r + nrows_x;
    if (seq1 > p.n_seq) seq1 = p.n_seq;
    const int nrows = seq1 > seq0r ? seq1 - seq0r : 0;
    int lane_h = lane;                                                    // (opaque per round: what is derived from it is not to be hoisted out of the round loop)
    asm volatile("" : "+v"(lane_h));
    // the round's row tables, for all eight waves (behind the previous round's closing barrier: nobody reads the old ones any more)
    {
      int seq = seq0r + lane_h;
      if (seq >= p.n_seq) seq = p.n_seq - 1;
      const int r0 = (int)((seq / p.inner) * p.outer + (seq % p.inner));
      rowtab[lane_h] = r0;
      xrow[lane_h] = lane_h < nrows ? (unsigned)r0 * (unsigned)((int)p.ldx * 2) : 0xFFFFF000u;
      for (int tt = lane_h; tt < XTHR; tt += 64) {                          // (the h piece of working thread tt: row tt / 7, units 8 (tt % 7) ..)
        const int sr = tt / (XUW * 2 / 16), sc = tt - sr * (XUW * 2 / 16), ucol = j * XUW + sc * 8;
        int sq = seq0r + sr;
        if (sq >= p.n_seq) sq = p.n_seq - 1;
        const int rr = (int)((sq / p.inner) * p.outer + (sq % p.inner));
        hrow[tt] = (sr < nrows && ucol < H) ? ((unsigned)rr * (unsigned)ldh_i + (unsigned)(hcol_i + ucol)) * 2u : 0xFFFFF000u;
      }
    }
    __syncthreads();                                                      // the round's row tables are written
    fetch4(0, (dir ? p.seq_len - 1 : 0) * stride_i, lane_h);                 // x_0 -> tile 0 (every wave its rows; B0 below)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NG = (NT * 16 * GPC + 63) / 64, NC = (NT * 16 * (XUW * 4 / 16) + 63) / 64;     // 30 gates pieces, 14 c pieces per lane and step (NT = 4)
    constexpr int GC = XUW * 8 / 16, CC = XUW * 4 / 16;
    unsigned og[NG], oc[NC];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int idx = lane_h + i * 64, row = idx / GPC, cc = idx - row * GPC;
      og[i] = (row < nrows && cc < GC && cc * 2 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldg_i + (unsigned)(gcol_i + (j * XUW + cc * 2) * 4)) * 2u : COOB;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int idx = lane_h + i * 64, row = idx / CC, cc = idx - row * CC;
      oc[i] = (row < nrows && cc * 4 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldc_i + (unsigned)(hcol_i + j * XUW + cc * 4)) * 4u : COOB;
    }
    uint4 vg[NG], vc[NC];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // B0: x_0 is in tile 0
    // x_t[192 .. 199] (16 bytes per row: bytes 384 .. 399 of the x row) -> chunk 49 of the h tile's row, where the recurrent product's last k-slab picks
    // it up (XNSP); lane = row.  The h tile's chunk 49 is touched by nobody else: the gather writes chunks 0 .. 48.
    auto copy_x49 = [&](int par) {
      const uint4 v = *reinterpret_cast<const uint4*>(gstage0 + par * (XROWS * GP) + lane * GP + XNSP * 64);
      *reinterpret_cast<uint4*>(htile + lane * pitch + (Hp - 32 + 8) * 2) = v;
    };
    copy_x49(0);
    // Between barrier 1 and barrier 2 (the working waves multiply and update the cells) the helper issues its four DMAs of x_{t+1} and waits for them;
    // behind barrier 2 (the working waves publish h_t, project x_{t+1}, gather) it reads the step's staged gate activations and c_t into registers and
    // stores them - the tile is free again when it arrives at the next barrier 1, which is when x_{t+2} may overwrite it.
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
    for (int step = 0; step < p.seq_len; ++step) {
      const int t = dir ? (p.seq_len - 1 - step) : step;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (chunk 49 is written)
      XST(8);
      __builtin_amdgcn_s_barrier();                                       // barrier 1 of the step
      XST(9);
      if (step + 1 < p.seq_len) fetch4((step + 1) & 1, (dir ? t - 1 : t + 1) * stride_i, lane);
      XST(10);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // its rows of x_{t+1} have landed (and the previous step's stores are done:
      XST(11);                                                            //  their data registers may be rewritten)
      __builtin_amdgcn_s_barrier();                                       // barrier 2 of the step
      copy_x49((step + 1) & 1);                                           // (x_{t+1} is whole behind barrier 2; read by the MFMAs behind barrier 1 of step t + 1)
      if (p.save) {
#if XHSLEEP > 0
        __builtin_amdgcn_s_sleep(XHSLEEP);                                // (the working waves' publication of h_t first: one store the cluster waits for)
#endif
        read_pieces(step & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        toff_st = t * stride_i;
        store_pieces();
      }
    }
    __builtin_amdgcn_s_barrier();                                         // end of the round: tables and tiles may be rewritten (the last step's pieces are in registers)
    }
    return;
  }
  // resident weight fragments of the working waves: loaded once, kept over every round
  uint4 breg[XQ][XNSH], wreg[XQ][XNSP];
  const int lu0 = w * XQ * 4 + lr;                                        // unit index inside the workgroup of quad 0 (quad 1: + 4); H % 56 == 0: all valid
  (void)lu0;
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
    const int qd = j * (XW * XQ) + w * XQ + q;
    const char* sh = reinterpret_cast<const char*>(p.whhq) + (((long)dir * nq + qd) * XNSH) * 1024 + lane * 16;
    const char* sx = reinterpret_cast<const char*>(p.wihq) + (((long)dir * nq + qd) * XNSX) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < XNSH; ++ks) breg[q][ks] = *reinterpret_cast<const uint4*>(sh + ks * 1024);
#pragma unroll
    for (int ks = 0; ks < XNSP; ++ks) wreg[q][ks] = *reinterpret_cast<const uint4*>(sx + ks * 1024);
    // the last slab of W_hh holds k = 384 .. 415, of which 392 .. 415 are padding (zero columns): lanes lr = 1 (k = 392 .. 399) take the seventh slab
    // of W_ih's lr = 0 lanes (input channels 192 .. 199, of which 196 .. 199 are padding) - the h tile's chunk 49 carries x_t[192 .. 199] (helper wave)
    const uint4 w6 = *reinterpret_cast<const uint4*>(sx + (XNSX - 1) * 1024);
    const int src = ((lane - 16) & 63) * 4;
    const uint4 w6s = make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w6.x), (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w6.y),
                                 (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w6.z), (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w6.w));
    if (lr == 1) breg[q][XNSH - 1] = w6s;
  }
  // The resident weight fragments are USED once here, in front of the time loop: the compiler's wait-count bookkeeping then knows their loads are done.
  // Without this the loads of the last fragments count as outstanding at the loop's entry, the merge with the back edge keeps that state for every
  // iteration, and the first MFMAs that read those registers carry `s_waitcnt vmcnt(4) ... vmcnt(0)` - which, from the second step on, wait for the
  // seven loads of the h GATHER issued just before: the projection ran BEHIND the gather's round trip instead of under it (found in the ISA after the
  // stamps put 2,900 - 3,250 cycles on a projection whose MFMAs need 1,500).
#pragma unroll
  for (int q = 0; q < XQ; ++q) {
#pragma unroll
    for (int ks = 0; ks < XNSH; ++ks) asm volatile("" :: "v"(breg[q][ks].x), "v"(breg[q][ks].y), "v"(breg[q][ks].z), "v"(breg[q][ks].w));
#pragma unroll
    for (int ks = 0; ks < XNSP; ++ks) asm volatile("" :: "v"(wreg[q][ks].x), "v"(wreg[q][ks].y), "v"(wreg[q][ks].z), "v"(wreg[q][ks].w));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bool dead = false;
  for (int rnd = 0; rnd < nrounds; ++rnd) {
  // ---- the round's sequences (their row tables are the HELPER wave's work: the integer divisions need two dozen registers, and here 160 hold weights);
  // the thread id is made opaque per round: left visible, everything derived from it is hoisted out of the round loop and stays live over every step
  int tid_o = tid;
  asm volatile("" : "+v"(tid_o));
  seq0 = seq0_first + rnd * rstride;
  int seq1 = seq0 + nrows_x;
  if (seq1 > p.n_seq) seq1 = p.n_seq;
  const int nrows = seq1 > seq0 ? seq1 - seq0 : 0;
  __syncthreads();                                                        // the round's row tables are written
  fetch4(0, (dir ? p.seq_len - 1 : 0) * stride_i, tid_o & 63);                 // x_0 -> tile 0 (every wave its rows; B0 below)


  // ================= working waves =================
  // operands swapped as in lstm_cluster.hip: A = the resident weight fragment (rows = the quad's 16 gate columns), B = the h / x fragment (columns =
  // sequences): lane (lr, lc) then holds the FOUR GATES of unit lr of the quad for sequence lc of the row tile
  // c_{t-1} is read back from the previous step's c staging tile (LDS, written every step), not carried in registers: the budget is 256 and the
  // resident weights take 160
  for (int i = tid_o; i < 2 * XROWS * XUW; i += XTHR) reinterpret_cast<float*>(cstage0)[i] = 0.f;

  // exchange planes: rows at the LDS tile's pitch (864 B), so that a chunk sits at the same byte offset of the plane AND of the tile, and a chunk walk whose
  // offsets are the lane's base + a compile-time multiple of the pass: pass i of thread tid < 441 takes chunk tid % 49 of row 9 i + tid / 49 (nine rows of
  // 49 data chunks per pass, 7 passes = 63 rows), the seven threads left over (441 .. 447, all in wave 6) share row 63, seven chunks per pass.  One per-lane
  // offset, everything else an instruction's immediate or a scalar (round 5: the step's vector ALU work, not its memory, is what this kernel waits for -
  // two working waves per SIMD issued ~730 vector instructions per step each, 40 % of them integer address arithmetic)
  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * p.rows_pad * pitch);
  const unsigned cl_bytes = (unsigned)((long)clx * p.rows_pad * pitch);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(2u * plane_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs2 = __builtin_amdgcn_make_buffer_rsrc(H2 ? p.hout2 : p.hout, 0, (int)p.h_bytes, 0x00020000);
  constexpr int HL = 7, HCH = 49, HRP = 9;                                // chunks per working thread, data chunks per row, rows per pass
  static_assert(XTHR == HRP * HCH + 7 && HRP * HL + 1 == XROWS && HL * 7 == HCH, "chunk walk below");
  const bool tailw = w == XW - 1;                                         // the wave with the seven left-over threads (uniform)
  const bool tailt = tid >= HRP * HCH;
  constexpr int GCORR = HRP * pitch - 7 * 16;                             // their pass stride is 7 chunks instead of 9 rows
  // Rows are published and gathered in whole passes: nact = ceil(nrows / 9) passes of nine rows (a cluster with fewer than 64 sequences - the mixed
  // clusters of the XCD-aware formation have 32 - moves 36 rows, not 64; the rows past its last sequence compute on zero inputs and are stored nowhere),
  // so that "which chunks does this thread wait for" is a scalar, not a per-lane mask.  With fewer than 64 rows the seven left-over threads
  // duplicate threads 0 .. 6 (same loads, same LDS writes) instead of taking row 63.
  constexpr int HLt = NT == 4 ? HL : (NT * 16 + HRP - 1) / HRP;           // passes an instance of fewer row tiles can need (2 / 4)
  const int nact = nrows >= XROWS ? HL : (nrows + HRP - 1) / HRP;
  const bool full = nrows >= XROWS;
  (void)tailt;
  constexpr unsigned TAGM = 0x40004000u;                                  // bit 14 of both 16-bit halves: clear in |h| <= 1 (bf16 and f16)
  // the one (row, 16-byte piece) of the staged h tile this thread publishes / stores per step: 64 rows x 7 pieces = 448 = one per working thread
  constexpr int SC = XUW * 2 / 16;
  const int st_row = tid / SC, st_cc = tid - st_row * SC;
  // LDS regions as byte offsets from smem
  constexpr int H0 = XROWS * pitch, HS = XROWS * XUW * 2;                  // h staging [2]
  constexpr int G0 = H0 + 2 * HS, GS = XROWS * GP;                         // x / gates tiles [2]
  constexpr int C0 = G0 + 2 * GS, CS = XROWS * XUW * 4;                    // c staging [2]
  constexpr int B0 = C0 + 2 * CS;                                          // bias
  // per-lane byte offsets, derived inside the step from the (opaque) lane id: lane (lr, lc), unit lu0 = w * 8 + lr of the workgroup (quad 1: + 4)
  //   A fragment of the h tile   lc * pitch + 16 lr              + rt * 16 * pitch + ks * 64
  //   A fragment of the x tile   G0 + lc * GP + 16 lr            + par * GS + rt * 16 * GP + ks * 64
  //   c staging of (lc, lu0)     C0 + lc * XUW * 4 + lu0 * 4     + par * CS + rt * 16 * XUW * 4 + q * 16      (h staging: H0 + half of the lane part)
  //   gate activations           G0 + lc * GP + lu0 * 8          + par * GS + rt * 16 * GP + q * 32
  //   bias of unit lu0           B0 + lu0 * 16                   + q * 64
  //   gather chunk / staged piece  tid * 16
  int toff_d = 0;
  bool have_d = false;
  const int gs0 = rnd * p.seq_len, gs_end = nrounds * p.seq_len;        // the hand-off's step counter runs on across rounds
  // (returns the registers the stores read: a 16-byte buffer store fetches its data some time AFTER it has issued, and an LDS read that the compiler
  //  places a few instructions behind it into the same registers can land first - seen here as the first dword of a piece replaced in the waves that
  //  issue last, and in lstm_cluster.hip's helper waves in round 4; the caller keeps the registers occupied until the MFMA block behind has issued)
  auto deferred_hout = [&](int par, int tv, uint4& keep0, uint4& keep1) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned dvo_h = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(hrow) + (tv >> 2));      // (a table, not a register held for 401 steps)
    const uint4 v = *reinterpret_cast<const uint4*>(smem + H0 + par * HS + tv);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_hs, (int)dvo_h, toff_d * ldh_i * 2, 0);
    keep0 = v;
    if constexpr (H2) {
      float a0, a1, a2, a3, a4, a5, a6, a7;
      unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
      const uint4 vb = make_uint4(pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7));
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{vb.x, vb.y, vb.z, vb.w}, rs_hs2, (int)dvo_h, toff_d * ldh_i * 2, 0);
      keep1 = vb;
    }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                           // B0: x_0 is in tile 0 (every wave fetched its rows)

  int lane_v = lane;
  volatile __attribute__((address_space(3))) unsigned* dead_lds = (volatile __attribute__((address_space(3))) unsigned*)deadflag;
  for (int step = 0; step < p.seq_len; ++step) {
    // The lane's indices are made opaque per step (and again per phase): left visible as loop invariants the compiler hoists every address derived
    // from them out of the time loop (~60 registers), the 256-register budget (160 of it resident weights) turns those into scratch, and a scratch
    // reload is a vector memory operation IN FRONT of which the in-order vmcnt queue waits for every gather load issued before it (the gather ran as
    // serial round trips).  Each phase derives its few base offsets from the lane id (a handful of integer instructions); everything else is an
    // instruction's immediate or a scalar.  (In place: no second copy of the registers stays live.)
    XST(0);
    asm volatile("" : "+v"(lane_v));
    // the thread's chunk walk base: (tid / 49) * pitch + (tid % 49) * 16 = 16 tid + (pitch - 49 * 16) * (tid / 49), tid / 49 = tid * 1338 >> 16 for tid < 448;
    // the seven left-over threads (wave 6, lanes 57 .. 63): row 63 of a full cluster, else they duplicate threads 0 .. 6
    int gb;
    {
      const int tid_v = w * 64 + lane_v;
      gb = tid_v * 16 + (pitch - HCH * 16) * ((tid_v * 1338) >> 16);
      if (tailw) gb = lane_v >= 64 - 7 ? (full ? (XROWS - 1) * pitch : 0) + (lane_v - (64 - 7)) * 16 : gb;
    }
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const int par = step & 1;
    const int gs = gs0 + step;
    const unsigned pprev = (unsigned)((gs + 1) & 1), pcur = (unsigned)(gs & 1);
    const unsigned tag_cur = (((unsigned)gs >> 1) & 1u) ^ 1u;
    const unsigned tag_prev = (((unsigned)(gs - 1) >> 1) & 1u) ^ 1u;
    const int xg = G0 + par * GS + (lane_v & 15) * GP + (lane_v >> 4) * 16;
    const int bq = B0 + w * (XQ * 64) + (lane_v >> 4) * 16;
    // ---- 0. x_t W_ih^T + b (independent of h) + 1. the h gather.  The gather's loads are issued in front of the projection.
    // (the projection's sums wait for the gather as 16-bit pairs of the operand format - what the two-kernel form stores in gx - : 16 registers
    //  instead of 32 across barrier 1)
#if XPROJ_F32
    f32x4_t accp[NT][XQ];
#else
    uint2 accp[NT][XQ];
#endif
    {
      // the thread's chunks of the cluster's h_{t-1}, all of them in flight at once.  Which chunks are still awaited is kept per WAVE (a scalar mask:
      // a chunk index is re-requested for the whole wave while any of its lanes misses a tag - re-reading a piece that has arrived is harmless, the plane
      // is not rewritten before every member has passed this step), so the loop carries no per-lane bookkeeping
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      uint4 hn[HLt];
      bool live = gs > 0 && !dead && nact > 0;                            // (dead: this wave has seen a hand-off time out - it stops waiting;
                                                                          //  nact == 0: a cluster the XCD-aware formation left without sequences)
#ifdef XABL_NO_GATHER
      live = false;
#endif
      const unsigned sbase = pprev * plane_bytes + cl_bytes;
      const int gcorr = (tailw && full && lane_v >= 64 - 7) ? GCORR : 0;
      // (all seven loads of a round are unconditional: a pass past the cluster's last row - clusters with fewer than 64 sequences - re-reads pass 0,
      //  a scalar select of the offset; per-chunk branches around the loads made the compiler keep two generations of the 28 registers alive)
      auto load_all = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < HLt; ++i) {
          const int so = (int)(sbase + (unsigned)(i < nact ? HRP * pitch * i : 0));
          const u32x4 r = tailw ? __builtin_amdgcn_raw_buffer_load_b128(rs, i < nact ? gb - i * gcorr : gb, so, 16)
                                : __builtin_amdgcn_raw_buffer_load_b128(rs, gb, so, 16);
          hn[i] = make_uint4(r[0], r[1], r[2], r[3]);
        }
      };
      if (live) load_all();
      else {
#pragma unroll
        for (int i = 0; i < HLt; ++i) hn[i] = make_uint4(0, 0, 0, 0);
      }
      unsigned pendu = live ? 1u : 0u;
      // a chunk has arrived when all eight tags are the previous step's: AND / OR trees over the round's chunks, one ballot per round
      auto check = [&]() __attribute__((always_inline)) {
        unsigned acc_and = hn[0].x & hn[0].y & hn[0].z & hn[0].w, acc_or = hn[0].x | hn[0].y | hn[0].z | hn[0].w;
#pragma unroll
        for (int i = 1; i < HLt; ++i) {
          acc_and &= hn[i].x & hn[i].y & hn[i].z & hn[i].w;
          acc_or |= hn[i].x | hn[i].y | hn[i].z | hn[i].w;
        }
        const bool miss = tag_prev ? ((acc_and & TAGM) != TAGM) : ((acc_or & TAGM) != 0u);
        if (__builtin_amdgcn_ballot_w64(miss) == 0ull) pendu = 0u;
      };
      auto reissue = [&]() __attribute__((always_inline)) { load_all(); };
      // ---- the projection, behind the loads: A fragments of the x tile (4 row tiles x 7 k-slabs) read XPD ahead through a rotating set of registers.
      // Between row tiles the wave looks at what has arrived and asks again for what has not: the members of a cluster publish within a few hundred
      // cycles of each other, so the first request usually comes back with the old tags, and a wave that only looked again after the whole projection
      // (4,100 cycles, in-kernel stamps) found out 2,600 cycles late.
#if XPROJ_RING
      // round 6: the A fragments of the x tile (4 row tiles x 6 k-slabs) through ONE rotating set of XPD registers across the row tiles, as the h tile's in
      // phase 2: in groups of XPD per row tile (below) every group waited out an LDS round trip - 2,800 cycles for 768 of MFMA per wave, and the gather
      // is back after ~2,050
      uint4 pa[XPD];
      auto prd = [&](int idx) __attribute__((always_inline)) {
#ifndef XABL_NO_AREAD
        return *reinterpret_cast<const uint4*>(smem + xg + (idx / XNSP) * 16 * GP + (idx % XNSP) * 64);
#else
        return make_uint4(xg, idx, 0, 0);
#endif
      };
#pragma unroll
      for (int i = 0; i < XPD; ++i) pa[i] = prd(i);
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        f32x4_t accx[XQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) accx[q] = *reinterpret_cast<const f32x4_t*>(smem + bq + q * 64);
#pragma unroll
        for (int k = 0; k < XNSP; ++k) {
          const int idx = rt * XNSP + k;
          const uint4 a = pa[idx % XPD];
          if (idx + XPD < NT * XNSP) pa[idx % XPD] = prd(idx + XPD);
#pragma unroll
#ifndef XABL_NO_PROJ
          for (int q = 0; q < XQ; ++q) accx[q] = mfma16<TI>(wreg[q][k], a, accx[q]);
#else
          for (int q = 0; q < XQ; ++q) accx[q][k & 3] += __uint_as_float(wreg[q][k].x ^ a.x);
#endif
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) accp[rt][q] = PROJ_KEEP(accx[q]);
#ifndef XNO_MIDPOLL
        if (((XMIDPOLL >> rt) & 1) && pendu) {
          check();
          if (pendu) reissue();
        }
#endif
      }
#else
      static_assert(NT == 4, "the grouped projection exists for the full instance only");
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        f32x4_t accx[XQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) accx[q] = *reinterpret_cast<const f32x4_t*>(smem + bq + q * 64);
#pragma unroll
        for (int k0 = 0; k0 < XNSP; k0 += XPD) {
          uint4 a[XPD];
#pragma unroll
          for (int i = 0; i < XPD; ++i)
#ifndef XABL_NO_AREAD
            if (k0 + i < XNSP) a[i] = *reinterpret_cast<const uint4*>(smem + xg + rt * 16 * GP + (k0 + i) * 64);
#else
            if (k0 + i < XNSP) a[i] = make_uint4(xg, rt, k0, i);
#endif
#pragma unroll
          for (int i = 0; i < XPD; ++i)
            if (k0 + i < XNSP) {
#pragma unroll
#ifndef XABL_NO_PROJ
              for (int q = 0; q < XQ; ++q) accx[q] = mfma16<TI>(wreg[q][k0 + i], a[i], accx[q]);
#else
              for (int q = 0; q < XQ; ++q) accx[q][i & 3] += __uint_as_float(wreg[q][k0 + i].x ^ a[i].x);
#endif
            }
        }
#pragma unroll
        for (int q = 0; q < XQ; ++q) accp[rt][q] = PROJ_KEEP(accx[q]);
#ifndef XNO_MIDPOLL
        if (((XMIDPOLL >> rt) & 1) && pendu) {
          check();
          if (pendu) reissue();
        }
#endif
      }
#endif
      XST(1);
      if (pendu) {
        unsigned spins = 0;
        while (true) {
          check();
          if (!pendu) break;
          __builtin_amdgcn_s_sleep(2);
          ++spins;
          if ((spins & 1023u) == 0u && dead_lds[0] != 0u) { dead = true; break; }      // another wave of the workgroup has given up
          if (spins > (1u << 20)) { atomicExch(p.err, 1u); dead_lds[0] = 1u; dead = true; break; }
          reissue();
        }
      }
      XST(2);
      // (tags of the other parity are zero bits: nothing to clear.  A later round's first step: the wait was for the PLANE, the round starts from h = 0 -
      //  the same AND with a mask of zeros; written as a separate `if (step == 0) hn = 0` the compiler spilled four weight fragments for the whole loop)
      const unsigned km = step == 0 ? 0u : ~TAGM;
      if (live && (tag_prev || step == 0)) {
#pragma unroll
        for (int i = 0; i < HLt; ++i) { hn[i].x &= km; hn[i].y &= km; hn[i].z &= km; hn[i].w &= km; }
      }
      if (!tailw) {
#pragma unroll
        for (int i = 0; i < HLt; ++i) *reinterpret_cast<uint4*>(smem + gb + HRP * pitch * i) = hn[i];      // (a dead pass wrote a copy of pass 0 into rows nobody stores)
      } else {
#pragma unroll
        for (int i = 0; i < HLt; ++i) *reinterpret_cast<uint4*>(smem + gb - i * gcorr + HRP * pitch * i) = hn[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    XST(3);
    __builtin_amdgcn_s_barrier();                                         // barrier 1: the h tile is whole; every wave has read x_t
    XST(4);
    asm volatile("" : "+v"(lane_v));
#if XFETCH_POS == 0
    if (step + 1 < p.seq_len) fetch4(par ^ 1, (dir ? t - 1 : t + 1) * stride_i, lane_v);      // this wave's rows of x_{t+1} (the tile's previous contents left before barrier 1)
#endif
    const int tv = w * 1024 + lane_v * 16;
    uint4 keep0 = make_uint4(0, 0, 0, 0), keep1 = keep0;
#ifndef XABL_NO_HOUT
    if (have_d) deferred_hout(par ^ 1, tv, keep0, keep1);                 // the previous step's h rows -> hout, under this step's MFMAs
#endif
    // ---- 2. + h_{t-1} W_hh^T, cell update - as a software pipeline inside the wave: the 26 MFMAs of row tile rt + 1 are issued among the vector
    // instructions of the cell update of row tile rt (one straight-line block: every quad and unit of this geometry is valid, the save switch is a
    // template parameter).
    const int a_off = (lane_v & 15) * pitch + (lane_v >> 4) * 16;
    const int cq = (lane_v & 15) * (XUW * 4) + (lane_v >> 4) * 4 + w * (XQ * 16);
    const int c_prev = cq + C0 + (par ^ 1) * CS, c_cur = cq + C0 + par * CS, h_cur = (cq >> 1) + H0 + par * HS;
    const int g_cur = G0 + par * GS + (lane_v & 15) * GP + (lane_v >> 4) * 8 + w * (XQ * 32);
    // The A fragments of the h tile (4 row tiles x 13 k-slabs) are read XAD ahead of their MFMAs through a rotating set of registers, across row-tile
    // boundaries: written as read -> MFMA -> MFMA the compiler kept that order and every k-slab waited out an LDS round trip (13 per row tile).
    uint4 ab[XAD];
    auto rd = [&](int idx) __attribute__((always_inline)) {
#ifndef XABL_NO_AREAD
      return *reinterpret_cast<const uint4*>(smem + a_off + (idx / XNSH) * 16 * pitch + (idx % XNSH) * 64);
#else
      return make_uint4(a_off, idx, 0, 0);
#endif
    };
#pragma unroll
    for (int i = 0; i < XAD; ++i) ab[i] = rd(i);
    auto mm = [&](int rt, f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        acc[q] = PROJ_TAKE(accp[rt][q]);
      }
#pragma unroll
      for (int ks = 0; ks < XNSH; ++ks) {
        const int idx = rt * XNSH + ks;
        const uint4 a = ab[idx % XAD];
        if (idx + XAD < 4 * XNSH) ab[idx % XAD] = rd(idx + XAD);
#pragma unroll
#ifndef XABL_NO_REC
        for (int q = 0; q < XQ; ++q) acc[q] = mfma16<TI>(breg[q][ks], a, acc[q]);
#else
        for (int q = 0; q < XQ; ++q) acc[q][ks & 3] += __uint_as_float(breg[q][ks].x ^ a.x);
#endif
      }
    };
    auto cell = [&](int rt, const f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < XQ; ++q) {
        const float cprev = *reinterpret_cast<const float*>(smem + c_prev + rt * 16 * XUW * 4 + q * 16);
#ifndef XABL_NO_CELL
        const float iv = sigmoidf_(acc[q][0]), fv = sigmoidf_(acc[q][1]), gv = tanhf_(acc[q][2]), ov = sigmoidf_(acc[q][3]);
        const float cv = __builtin_fmaf(fv, cprev, __fmul_rn(iv, gv));    // (spelled out: left to the compiler, the eight instances of a step and the
        const float hv = ov * tanhf_(cv);                                  //  template variants did not all contract the same way - 1 ulp apart)
#else
        const float iv = acc[q][0], fv = acc[q][1], gv = acc[q][2], ov = acc[q][3];
        const float cv = fv * cprev + iv * gv;
        const float hv = ov * cv;
#endif
        *reinterpret_cast<TI*>(smem + h_cur + rt * 16 * XUW * 2 + q * 8) = from_f32<TI>(hv);
        *reinterpret_cast<float*>(smem + c_cur + rt * 16 * XUW * 4 + q * 16) = cv;      // (also the next step's c_{t-1})
        if constexpr (SAVE) {
          uint2 gs;
          gs.x = pack2<bf16_t>(iv, fv);
          gs.y = pack2<bf16_t>(gv, ov);
          *reinterpret_cast<uint2*>(smem + g_cur + rt * 16 * GP + q * 32) = gs;
        }
      }
    };
#if XPIPE
    {
      // Round 6: ONE row tile ahead, interleaved by hand.  The cell update of row tile rt is cut into 13 stages (one per k-slab: a handful of vector
      // instructions each, the same operations in the same order as sigmoidf_ / tanhf_ - bit-identical results) and stage ks is issued right behind
      // the two MFMAs of k-slab ks of row tile rt + 1, with a scheduling fence between slabs.  Left to the compiler (the form below) a step's phase 2
      // is blocks of 18 - 26 MFMAs followed by blocks of 60 - 170 vector instructions: the two working waves of a SIMD run in lockstep behind
      // barrier 1, so both want the matrix pipe, then both want the vector ALU - 7,700 cycles for 3,300 of MFMA and 3,600 of VALU per SIMD.
      f32x4_t accS[2][XQ];
      float sx0[XQ], sx1[XQ], sx2[XQ], sx3[XQ], scp[XQ], scv[XQ], sec[XQ];
      auto mm_slab = [&](int rt, int ks, f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
        if (ks == 0) {
#pragma unroll
          for (int q = 0; q < XQ; ++q) {
            acc[q] = PROJ_TAKE(accp[rt][q]);
          }
        }
        const int idx = rt * XNSH + ks;
        const uint4 a = ab[idx % XAD];
        if (idx + XAD < NT * XNSH) ab[idx % XAD] = rd(idx + XAD);
#pragma unroll
        for (int q = 0; q < XQ; ++q) acc[q] = mfma16<TI>(breg[q][ks], a, acc[q]);
      };
      auto cell_stage = [&](int rt, int st, const f32x4_t (&acc)[XQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
          if (st == 0) { scp[q] = *reinterpret_cast<const float*>(smem + c_prev + rt * 16 * XUW * 4 + q * 16); sx0[q] = __expf(-acc[q][0]); }
          else if (st == 1) sx1[q] = __expf(-acc[q][1]);
          else if (st == 2) sx3[q] = __expf(-acc[q][3]);
          else if (st == 3) sx2[q] = __expf(2.0f * acc[q][2]);
          else if (st == 4) sx0[q] = __builtin_amdgcn_rcpf(1.0f + sx0[q]);                    // i
          else if (st == 5) sx1[q] = __builtin_amdgcn_rcpf(1.0f + sx1[q]);                    // f
          else if (st == 6) sx3[q] = __builtin_amdgcn_rcpf(1.0f + sx3[q]);                    // o
          else if (st == 7) sx2[q] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(sx2[q] + 1.0f);      // g
          else if (st == 8) {
            scv[q] = __builtin_fmaf(sx1[q], scp[q], __fmul_rn(sx0[q], sx2[q]));
            *reinterpret_cast<float*>(smem + c_cur + rt * 16 * XUW * 4 + q * 16) = scv[q];    // (also the next step's c_{t-1})
          } else if (st == 9) sec[q] = __expf(2.0f * scv[q]);
          else if (st == 10) sec[q] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(sec[q] + 1.0f);
          else if (st == 11) *reinterpret_cast<TI*>(smem + h_cur + rt * 16 * XUW * 2 + q * 8) = from_f32<TI>(sx3[q] * sec[q]);
          else if (st == 12) {
            if constexpr (SAVE) {
              uint2 gs_;
              gs_.x = pack2<bf16_t>(sx0[q], sx1[q]);
              gs_.y = pack2<bf16_t>(sx2[q], sx3[q]);
              *reinterpret_cast<uint2*>(smem + g_cur + rt * 16 * GP + q * 32) = gs_;
            }
          }
        }
      };
#pragma unroll
      for (int ks = 0; ks < XNSH; ++ks) mm_slab(0, ks, accS[0]);
#if XFETCH_POS == 1
      if (step + 1 < p.seq_len) fetch4(par ^ 1, (dir ? t - 1 : t + 1) * stride_i, lane_v);
#endif
#ifndef XNO_KEEP
      asm volatile("" :: "v"(keep0.x), "v"(keep0.y), "v"(keep0.z), "v"(keep0.w));      // (see deferred_hout)
      if constexpr (H2) asm volatile("" :: "v"(keep1.x), "v"(keep1.y), "v"(keep1.z), "v"(keep1.w));
#endif
      XST(12);
#pragma unroll
      for (int rt = 0; rt < NT - 1; ++rt) {
#pragma unroll
        for (int ks = 0; ks < XNSH; ++ks) {
          __builtin_amdgcn_sched_barrier(0);
          mm_slab(rt + 1, ks, accS[(rt + 1) & 1]);
          cell_stage(rt, ks, accS[rt & 1]);
        }
        if (rt == 0) XST(13);
        if (rt == 1) XST(14);
        if (rt == 2) XST(15);
      }
#pragma unroll
      for (int st = 0; st < 13; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        cell_stage(NT - 1, st, accS[(NT - 1) & 1]);
      }
    }
#else
    {
      static_assert(NT == 4, "the compiler-ordered phase 2 exists for the full instance only");
      f32x4_t accA[XQ], accB[XQ];
      mm(0, accA);
      mm(1, accB);
#ifndef XNO_KEEP
      asm volatile("" :: "v"(keep0.x), "v"(keep0.y), "v"(keep0.z), "v"(keep0.w));      // (see deferred_hout)
      if constexpr (H2) asm volatile("" :: "v"(keep1.x), "v"(keep1.y), "v"(keep1.z), "v"(keep1.w));
#endif
#if XFETCH_POS == 1
      if (step + 1 < p.seq_len) fetch4(par ^ 1, (dir ? t - 1 : t + 1) * stride_i, lane_v);
#endif
      cell(0, accA);
#ifdef XSCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {      // one A fragment read + its two MFMAs per ~10 vector instructions of the cell update
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
      }
#endif
      mm(2, accA);
      cell(1, accB);
#ifdef XSCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 1);
      }
#endif
      mm(3, accB);
      cell(2, accA);
#ifdef XSCHED
#pragma unroll
      for (int i = 0; i < XNSH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 2);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 2);
      }
#endif
      cell(3, accB);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");         // (its rows of x_{t+1} have landed)
    XST(5);
    __builtin_amdgcn_s_barrier();                                         // barrier 2
    XST(6);
    // ---- 3. h_t of this workgroup's units -> exchange buffer (tagged)
    if (gs + 1 < gs_end) {
      const unsigned tagv = tag_cur ? TAGM : 0u;
      const uint4 v = *reinterpret_cast<const uint4*>(smem + H0 + par * HS + tv);
      const uint4 vt = make_uint4(v.x | tagv, v.y | tagv, v.z | tagv, v.w | tagv);
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef XABL_NO_XSTORE
      // its place in the exchange plane: row tid / 7 at the tile's pitch, piece tid % 7 of this workgroup's seventh: 16 tid + (pitch - 112) (tid / 7) + 112 j,
      // tid / 7 = tid * 9363 >> 16 for tid < 448; rows past the last whole pass of a cluster with fewer than 64 sequences are not published
      const int srow = ((tv >> 4) * 9363) >> 16;
      unsigned xo = (unsigned)(tv + (pitch - XUW * 2) * srow + j * (XUW * 2));
      if (!full) xo = srow < nact * HRP ? xo : COOB;
      if (local) __builtin_amdgcn_raw_buffer_store_b128(u32x4{vt.x, vt.y, vt.z, vt.w}, rs, (int)xo, (int)(pcur * plane_bytes + cl_bytes), 0);
      else __builtin_amdgcn_raw_buffer_store_b128(u32x4{vt.x, vt.y, vt.z, vt.w}, rs, (int)xo, (int)(pcur * plane_bytes + cl_bytes), 16);
#else
      asm volatile("" :: "v"(vt.x));
#endif
    }
    XST(7);
    toff_d = toff;
    have_d = true;
  }
  {
    uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
    if (have_d) deferred_hout((p.seq_len + 1) & 1, w * 1024 + lane_v * 16, k0, k1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // end of the round: tables and tiles may be rewritten
    asm volatile("" :: "v"(k0.x), "v"(k0.y), "v"(k0.z), "v"(k0.w));      // (the store's data registers stay occupied until then, see deferred_hout)
    if constexpr (H2) asm volatile("" :: "v"(k1.x), "v"(k1.y), "v"(k1.z), "v"(k1.w));
  }
  }
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
         (size_t)XUW * 16 + XROWS * sizeof(int) + 32 * sizeof(int) + (XROWS + XTHR) * sizeof(unsigned);
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

#ifdef XSTAMP
extern "C" int urse_diag_clusterx_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xstamps), sizeof(unsigned long long) * 512 * 16);
}
#endif

extern "C" int urse_lstm_clusterx_supported(int N, int Np, int H, int Hp) {
  // the projection multiplies input channels 0 .. 199 (six slabs of W_ih + the lr = 0 lanes of the seventh, carried in chunk 49 of the h
  // tile): N in 201 .. 224 would silently lose channels; the 49-chunk gather, XNSH and XUW are written for H = 392 (7 workgroups x 56 units): a smaller
  // multiple of 56 would wait for chunks nobody publishes.  The model's shape is N = 196, H = 392.
  return (N > 0 && N <= 200 && Np == 224 && H == 392 && Hp == 416) ? 1 : 0;
}

// workspace / geometry query of the fused cluster forward: {C, ncl, rows_per_cluster, rows_pad, hx_elems (16-bit elements of the exchange planes at the LDS
// tile's pitch), n_counters, rounds}.  Up to ncl * 64 sequences per direction: urse_lstm_cluster_plan's clusters, one round.  More (the band path:
// 12,832 per direction): every co-resident cluster takes 64 sequences per round, rounds = ceil(n_seq / (ncl * 64)).
extern "C" int urse_lstm_clusterx_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && n_seq > 0 && reserved_cus >= 0, "urse_lstm_clusterx_plan: bad argument");
  URSE_CHECK_ARG(urse_lstm_clusterx_supported(196, 224, H, Hp), "urse_lstm_clusterx_plan: unsupported H=%d Hp=%d", H, Hp);
  int64_t one[6];
  int rc = urse_lstm_cluster_plan(H, Hp, n_seq < XROWS ? n_seq : XROWS, reserved_cus, one);      // (C, and whether a cluster fits beside the reservation at all)
  if (rc) return rc;
  const int ncl_max = (device_cu_count() - reserved_cus - 4) / 2 / (int)one[0];
  if (n_seq <= ncl_max * XROWS) {
    rc = urse_lstm_cluster_plan(H, Hp, n_seq, reserved_cus, plan);
    if (rc) return rc;
    plan[6] = 1;
  } else {
    plan[0] = one[0]; plan[1] = ncl_max; plan[2] = XROWS; plan[3] = XROWS; plan[5] = 2 * ncl_max;
    plan[6] = (n_seq + ncl_max * XROWS - 1) / (ncl_max * XROWS);
  }
  plan[4] = (int64_t)2 * 2 * plan[1] * plan[3] * (lds_frag_pitch(XNSH * 64) / 2);
  return URSE_OK;
}

extern "C" int urse_lstm_clusterx_hx_elems(int H, int Hp, int n_seq, int reserved_cus, int64_t* elems) {
  URSE_CHECK_ARG(elems, "urse_lstm_clusterx_hx_elems: null pointer");
  int64_t plan[7];
  const int rc = urse_lstm_clusterx_plan(H, Hp, n_seq, reserved_cus, plan);
  if (rc) return rc;
  *elems = plan[4];
  return URSE_OK;
}

extern "C" int urse_lstm_clusterx_fwd(const void* xn, int64_t ldx, const void* wihq, const float* bias, const void* whhq, void* gates, int64_t ldg,
                                      void* hout, int64_t ldh, float* c, void* hx, void* counters, void* err_flag, int N, int Np, int H, int Hp,
                                      int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save, int reserved_cus,
                                      int xcd_aware, int dtype, void* hout_bf16, void* stream) {
  URSE_CHECK_ARG(xn && wihq && bias && whhq && hout && hx && counters && err_flag && ((c && gates) || !save), "urse_lstm_clusterx_fwd: null pointer");
  URSE_CHECK_ARG(urse_lstm_clusterx_supported(N, Np, H, Hp), "urse_lstm_clusterx_fwd: unsupported N=%d Np=%d H=%d Hp=%d", N, Np, H, Hp);
  URSE_CHECK_ARG(dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_clusterx_fwd: operands are bf16 or f16 (dtype %d)", dtype);
  URSE_CHECK_ARG(!hout_bf16 || (dtype == URSE_F16 && ((uintptr_t)hout_bf16 % 16) == 0), "urse_lstm_clusterx_fwd: the bf16 copy of h goes with f16 operands only");
  int64_t plan[7];
  int rc = urse_lstm_clusterx_plan(H, Hp, n_seq, reserved_cus, plan);      // the same clusters as urse_lstm_cluster_fwd: 7 workgroups x 56 units, 64 sequences (per round)
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
  p.C = (int)plan[0]; p.ncl = (int)plan[1]; p.rows_per_cluster = (int)plan[2]; p.rows_pad = (int)plan[3]; p.rounds = (int)plan[6];
  hipStream_t st = (hipStream_t)stream;
  // the exchange planes start with every tag bit clear; their rows sit at the LDS tile's pitch: plan[4] * 27 / 26 elements (urse_lstm_clusterx_hx_elems)
  (void)hipMemsetAsync(hx, 0, (size_t)2 * 2 * plan[1] * plan[3] * lds_frag_pitch(XNSH * 64), st);
  p.xws = nullptr;
  if (xcd_aware && plan[5] >= 9) {
    (void)hipMemsetAsync(counters, 0, sizeof(unsigned) * plan[5], st);
    p.xws = (unsigned*)counters;
  }
  // row tiles per step: 64 sequences per cluster (and every launch in rounds) take the full instance; a plan of at most 16 / 32 per cluster - the time path of a small
  // batch: B <= 8 / 16 utterances at 48 kHz - the instances that gather, fetch, multiply and store one / two row tiles (URSE_CLUSTERX_NT = 1 .. 4 forces one that fits)
  // (the XCD-aware formation re-balances the sequences over the clusters that sit inside one XCD - two mixed clusters per direction without any is its usual case -,
  //  so the instance is chosen for n_seq over two clusters fewer; a formation that still needs more rows per cluster falls back to the plan's static clusters)
  const int rows_bound = (xcd_aware && plan[5] >= 9 && plan[1] > 2) ? (int)((n_seq + plan[1] - 3) / (plan[1] - 2)) : (int)plan[2];
  int nt = (plan[6] > 1 || rows_bound > 48) ? 4 : (rows_bound > 32 ? 3 : (rows_bound > 16 ? 2 : 1));
  if (const char* e = getenv("URSE_CLUSTERX_NT")) {
    const int f = atoi(e);
    if (f >= 1 && f <= 4 && f >= nt) nt = f;
  }
  const size_t lds = clusterx_lds();
  dim3 grid(p.C * p.ncl, 2), blk(XTHR + 64);
  note_launch(URSE_KV_LSTM_FWD_CLUSTERX);
#define URSE_CLX_GO(...) do { \
    static bool once_ = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_clusterx_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); \
    (void)once_; \
    hipLaunchKernelGGL((lstm_fwd_clusterx_kernel<__VA_ARGS__>), grid, blk, lds, st, p); } while (0)
#define URSE_CLX_NT(...) do { if (nt == 1) URSE_CLX_GO(__VA_ARGS__, 1); else if (nt == 2) URSE_CLX_GO(__VA_ARGS__, 2); else if (nt == 3) URSE_CLX_GO(__VA_ARGS__, 3); else URSE_CLX_GO(__VA_ARGS__, 4); } while (0)
  if (dtype == URSE_F16 && hout_bf16 && save) URSE_CLX_NT(f16_t, true, true);
  else if (dtype == URSE_F16 && save) URSE_CLX_NT(f16_t, false, true);
  else if (dtype == URSE_F16) URSE_CLX_NT(f16_t, false, false);
  else if (save) URSE_CLX_NT(bf16_t, false, true);
  else URSE_CLX_NT(bf16_t, false, false);
#undef URSE_CLX_NT
#undef URSE_CLX_GO
  URSE_CHECK_LAUNCH("urse_lstm_clusterx_fwd");
  return URSE_OK;
}
