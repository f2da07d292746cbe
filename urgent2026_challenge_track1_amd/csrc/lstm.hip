// Bidirectional LSTM recurrence (forward and backward-through-time) for the BSRNN dual-path blocks.
//
// The input projection x*W_ih^T + b (a plain big GEMM) is done by gemm_nt; these kernels run the
// sequential part.  Sequences are independent, so a workgroup owns 16*RT sequences of ONE direction
// for the whole time loop: h_{t-1} lives in LDS (MFMA A operand), c_t and the recurrent gradient live
// in registers in the MFMA C layout (lane = hidden unit, register = sequence), and the recurrent
// weights stream from L2 as MFMA B fragments every step.  There is no inter-workgroup communication
// and no grid barrier.  Gate order i,f,g,o and the two-bias convention follow nn.LSTM (cuDNN) as used
// by espnet2 BSRNN (twin: baseline_code/models/bsrnn_flowse.py:296-299 time path, :303-306 band path).
//
// Layouts chosen for the memory system: gate columns are interleaved per hidden unit
// (col = dir*4H + u*4 + gate) so a lane touches its four gates with ONE 8-byte access and 16 lanes cover
// a full 128-byte line; the recurrent weights are pre-packed in MFMA-fragment order (1 KiB per
// wave-instruction, fully coalesced) by urse_lstm_pack.
//
// row(s, t) = (s / inner) * outer + (s % inner) + t * stride   maps (sequence, step) to a row of the
// [B*T*K, .] channel-last activation matrices: time path inner=K, outer=T*K, stride=K; band path
// inner=1, outer=K, stride=1.
#include <type_traits>

#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct SeqMap {
  long inner, outer, stride;
  int n_seq, seq_len;
};

struct LstmFwdArgs {
  void* gx; long ldg;        // [M, ldg] T: gate pre-activations (both directions, 2*4H); overwritten with activations
  const void* whh;           // fragment-ordered [2][nut][nslab][4][64][16 B]
  void* hout; long ldh;      // [M, ldh] T: h (dir 0 cols [0,H), dir 1 cols [H,2H))
  void* hout2;               // f16 mode: the same h once more in bf16 (operand of the weight-gradient GEMMs; may be null)
  float* c;                  // [M, 2H] f32 cell state (saved when `save`)
  int H, Hp, save;
  int xcd;                   // xcd_dir_tile mapping
  SeqMap m;
};

struct LstmBwdArgs {
  const void* dh; long ldd;  // [M, ldd] T: gradient w.r.t. hout
  void* gates; long ldg;     // in: saved gate activations; out: gradient w.r.t. gate pre-activations
  const float* c;            // [M, 2H]
  const void* whhT;          // fragment-ordered [2][nut][nslabT][64][16 B]
  int H;
  int dbuf;                  // two LDS tiles (set by the launcher when they fit)
  int xcd;                   // xcd_dir_tile mapping
  SeqMap m;
};

template <typename T> struct Vec4;
template <> struct Vec4<bf16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ void unpack(raw v, float (&o)[4]) {
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  static __device__ __forceinline__ raw pack(const float (&i)[4]) {
    raw v;
    v.x = (unsigned)f32_to_bf16(i[0]) | ((unsigned)f32_to_bf16(i[1]) << 16);
    v.y = (unsigned)f32_to_bf16(i[2]) | ((unsigned)f32_to_bf16(i[3]) << 16);
    return v;
  }
};
template <> struct Vec4<f16_t> {
  typedef uint2 raw;
  static __device__ __forceinline__ void unpack(raw v, float (&o)[4]) { unpack2<f16_t>(v.x, o[0], o[1]); unpack2<f16_t>(v.y, o[2], o[3]); }
  static __device__ __forceinline__ raw pack(const float (&i)[4]) {
    raw v;
    v.x = pack2<f16_t>(i[0], i[1]);
    v.y = pack2<f16_t>(i[2], i[3]);
    return v;
  }
};
template <> struct Vec4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(raw v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ raw pack(const float (&i)[4]) { return make_float4(i[0], i[1], i[2], i[3]); }
};

template <typename T, int RT>
__device__ __forceinline__ void mma_slab(const uint4 (&a)[RT], const uint4& b, f32x4_t (&acc)[RT]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = mfma16<T>(a[rt], b, acc[rt]);
  } else {
    const float bf[4] = {__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float af[4] = {__uint_as_float(a[rt].x), __uint_as_float(a[rt].y), __uint_as_float(a[rt].z),
                           __uint_as_float(a[rt].w)};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[rt], 0, 0, 0);
    }
  }
}

// NW = waves per workgroup (16: most memory-level parallelism, 128 VGPRs; 8: 256 VGPRs for bigger row tiles)
template <typename T, int RT, int MAXUT, int NW>
__global__ void __launch_bounds__(NW * 64) lstm_fwd_kernel(LstmFwdArgs p) {
  constexpr int NTHR = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T), R = 16 * RT;
  typedef typename Vec4<T>::raw V4;
  // the saved gate activations feed the BPTT, whose operands are bf16 in the f16 forward mode too (gradients need the range)
  typedef typename std::conditional<__is_same(T, f16_t), bf16_t, T>::type TS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  int dir, tile_;
  xcd_dir_tile(p.xcd, dir, tile_);
  const int s0 = tile_ * R;
  const int H = p.H, Hp = p.Hp, nut = (H + 15) >> 4;
  const int pitch = lds_frag_pitch(Hp * ES);
  for (int i = tid; i < 2 * R * pitch / 4; i += NTHR) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  float cst[MAXUT][RT][4];
#pragma unroll
  for (int a = 0; a < MAXUT; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) cst[a][b][c] = 0.f;
  int rowbase[RT][4];
  bool rvalid[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      rvalid[rt][r] = seq < p.m.n_seq;
      if (seq >= p.m.n_seq) seq = p.m.n_seq - 1;
      rowbase[rt][r] = (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner));
    }
  const int nslab = Hp * ES / 64;
  // fragment-ordered weights: block(ut, ks, g) = 1 KiB, lane-linear
  const char* whh = reinterpret_cast<const char*>(p.whh) + ((long)dir * nut * nslab * 4) * 1024 + lane * 16;
  T* gx = reinterpret_cast<T*>(p.gx);
  T* hout = reinterpret_cast<T*>(p.hout);
  const long gcol0 = (long)dir * 4 * H;
  __syncthreads();

  for (int step = 0; step < p.m.seq_len; ++step) {
    const int t = dir ? (p.m.seq_len - 1 - step) : step;
    char* hc = smem + (step & 1) * R * pitch;
    char* hn = smem + ((step & 1) ^ 1) * R * pitch;
    const long toff = (long)t * p.m.stride;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        const int u = ut * 16 + lc;
        const bool uvalid = u < H;
        const int uc = uvalid ? u : H - 1;
        // gate pre-activations of this lane's (row, unit) pairs: independent of the recurrence, issued first
        V4 gxv[RT][4];
#ifndef ABL_NO_PW
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            gxv[rt][r] = *reinterpret_cast<const V4*>(gx + ((long)rowbase[rt][r] + toff) * p.ldg + gcol0 + uc * 4);
#endif
        f32x4_t acc[4][RT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[g][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* wr = whh + ((long)ut * nslab * 4) * 1024;
        const char* ar = hc + lc * pitch + 16 * lr;
        // the weight stream (L2 -> registers) is the bottleneck: keep KB slabs x 4 gates of fragment loads in flight
#ifndef URSE_FWD_KB
#define URSE_FWD_KB 3
#endif
        constexpr int KB = (RT == 1 && sizeof(T) == 2) ? URSE_FWD_KB : 3;
        for (int k0 = 0; k0 < nslab; k0 += KB) {
          uint4 b[KB][4];
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (k0 + i < nslab) ? k0 + i : nslab - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) b[i][g] = *reinterpret_cast<const uint4*>(wr + (ks * 4 + g) * 1024);
          }
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            if (k0 + i < nslab) {
              uint4 a[RT];
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
                a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + (k0 + i) * 64);
#pragma unroll
              for (int g = 0; g < 4; ++g) mma_slab<T, RT>(a, b[i][g], acc[g]);
            }
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long row = (long)rowbase[rt][r] + toff;
            float pre[4] = {0.f, 0.f, 0.f, 0.f};
#ifndef ABL_NO_PW
            Vec4<T>::unpack(gxv[rt][r], pre);
#endif
            const float iv = sigmoidf_(acc[0][rt][r] + pre[0]), fv = sigmoidf_(acc[1][rt][r] + pre[1]);
            const float gv = tanhf_(acc[2][rt][r] + pre[2]), ov = sigmoidf_(acc[3][rt][r] + pre[3]);
            const float cv = fv * cst[ui][rt][r] + iv * gv;
            cst[ui][rt][r] = cv;
            const float hv = uvalid ? ov * tanhf_(cv) : 0.f;
            const T hT = from_f32<T>(hv);
            *reinterpret_cast<T*>(hn + (rt * 16 + lr * 4 + r) * pitch + u * ES) = hT;
#ifdef ABL_NO_PW
            if (rvalid[rt][r] && uvalid && step == p.m.seq_len - 1) {
#else
            if (rvalid[rt][r] && uvalid) {
#endif
              hout[row * p.ldh + (long)dir * H + u] = hT;
              if constexpr (__is_same(T, f16_t)) {
                if (p.hout2) reinterpret_cast<bf16_t*>(p.hout2)[row * p.ldh + (long)dir * H + u] = f32_to_bf16(to_f32<T>(hT));      // (the f16 value rounded once more, as the other forward kernels' copies)
              }
              if (p.save) {
                const float act[4] = {iv, fv, gv, ov};
                *reinterpret_cast<V4*>(gx + row * p.ldg + gcol0 + u * 4) = Vec4<TS>::pack(act);
                p.c[row * 2 * H + (long)dir * H + u] = cv;
              }
            }
          }
      }
    }
    __syncthreads();
  }
}

// wave priority of the BPTT kernels (measured neutral against the wgrad GEMMs on the second queue: 180.3 vs 179.9 ms/step)
#ifndef URSE_BWD_PRIO
#define URSE_BWD_PRIO 0
#endif
// HC: hidden size known at compile time (0 = run-time p.H): tile pitch, slab count and unit-tile count fold into constants
// HPC: only the LDS tile pitch is folded (the 32-row variant spills when its loop bounds become constants too)
// PF: software pipeline over the steps.  The step's inputs (saved gates, c_{t-1}, dh: 14 bytes per (row, unit), from HBM) do not
// depend on the recurrence.  PF = 0 loads them at the top of the step, one unit tile at a time (registers: the 16-wave
// geometry is capped at 128), i.e. MAXUT dependent round trips to HBM per step in front of the weight pass.  PF = 1 issues the
// loads of step t + 1 for ALL the wave's unit tiles right after the cell update of step t, in front of the step's weight pass:
// they are older than every weight fragment in the wave's in-order vmcnt queue, so they have landed when the first fragment
// has, and the next step starts on registers.  Same arithmetic in the same order: bit-identical results.
// STG: the step's gate gradients leave through the LDS tile (which holds them anyway) as 16-byte pieces along the rows, after the barrier,
// instead of 8 bytes per lane and unit from the cell phase (32 store instructions per lane and step in the 32-row geometry): what took
// 1.2 us per step off the N-split kernel (lstm_nsplit.hip).  bf16, single direction segment of 4H columns a multiple of 8.
// KH = 2 (f32 operands at H > 624: the [16][4H] f32 tile of the flow model's H = 768 is 196 KB, more than a CU's LDS): the tile holds one HALF of
// the gate-gradient columns at a time; a wave keeps the gradients of its units in registers, the recurrent product runs half by half over the
// matching half of the k range into the same accumulators (same k order as one pass: bit-identical to a tile that would fit).
#ifdef BWSTAMP      // timing diagnostics (scripts/stamps.py): shader-clock stamps of wave 0 of workgroup BWSTAMP, [step][16]: 0 top of the step, 1 .. MAXUT after the cell
                    // update of unit tile ui (its inputs loaded, the gradients in the LDS tile), 5 at the barrier, 6 behind it, 7 the staged stores issued,
                    // 8 .. 8 + MAXUT - 1 after the weight pass of unit tile ui
__device__ unsigned long long g_bwstamps[512 * 16];
#define BWST(slot) do { if (stamp_on && step < 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_bwstamps[step * 16 + (slot)] = t_; } } while (0)
#else
#define BWST(slot) do { } while (0)
#endif
// one LDS-DMA wave-instruction (64 lanes x 16 B, lane-linear at the LDS byte address dst; m0 declared clobbered, as in gemm.hip)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void bw_glds16(const char* gsrc, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop
template <typename T, int RT, int MAXUT, int NW, int HC = 0, int HPC = 0, int PF = 0, int STG = 0, int KH = 1>
__global__ void __launch_bounds__(NW * 64) lstm_bwd_kernel(LstmBwdArgs p) {
  static_assert(KH == 1 || (KH == 2 && !HC && !HPC && !PF && !STG), "the half-tile form exists for the plain variant only");
#ifdef BWSTAMP
  const bool stamp_on = blockIdx.x == BWSTAMP && (threadIdx.x >> 6) == 0;
#endif
#if URSE_BWD_PRIO
  __builtin_amdgcn_s_setprio(URSE_BWD_PRIO);   // the BPTT is on the step's critical path; the wgrad GEMMs it shares CUs with are not
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T), R = 16 * RT;
  typedef typename Vec4<T>::raw V4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  int dir, tile_;
  xcd_dir_tile(p.xcd, dir, tile_);
  const int s0 = tile_ * R;
  const int H = HC ? HC : p.H, nut = (H + 15) >> 4, G4 = 4 * H;
  const int pitch = HPC ? lds_frag_pitch(4 * HPC * ES) : lds_frag_pitch(G4 / KH * ES);
  const int nbuf = (p.dbuf && KH == 1) ? 2 : 1;         // double-buffered dgates tile: one barrier per step
  float dcs[MAXUT][RT][4], dhr[MAXUT][RT][4], ccur[MAXUT][RT][4];
  int rowbase[RT][4];                      // negative: sequence beyond n_seq (clamped, never stored)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      const bool ok = seq < p.m.n_seq;
      if (!ok) seq = p.m.n_seq - 1;
      const int rb = (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner));
      rowbase[rt][r] = ok ? rb : -rb - 1;
    }
  auto rowb = [&](int rt, int r) -> long { return rowbase[rt][r] >= 0 ? rowbase[rt][r] : -(rowbase[rt][r] + 1); };
  if constexpr (STG) {                       // row of each of the workgroup's sequences at t = 0 (-1 beyond n_seq), behind the tile(s)
    int* rowtab = reinterpret_cast<int*>(smem + nbuf * R * pitch);
    if (tid < R) {
      const int seq = s0 + tid;
      rowtab[tid] = seq < p.m.n_seq ? (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner)) : -1;
    }
  }
  const int nslab = G4 * ES / 64;
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * nut * nslab) * 1024 + lane * 16;
  const T* dh = reinterpret_cast<const T*>(p.dh);
  T* gates = reinterpret_cast<T*>(p.gates);
  // 32-bit row indices / leading dimensions (checked on the host): an address costs one v_mad_i64_i32
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.m.stride;
#ifdef BABL_ALIGNED    // timing diagnostic (wrong results): both directions read / write direction 0's columns, whose rows are 128-byte aligned
  const int gcol_i = 0, hcol_i = 0, prev_i = dir ? stride_i : -stride_i;
#else
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
#endif

  {
    const long toff0 = (long)(dir ? 0 : p.m.seq_len - 1) * p.m.stride;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int u = (w + NW * ui) * 16 + lc;
      const int uc = u < H ? u : H - 1;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dcs[ui][rt][r] = 0.f;
          dhr[ui][rt][r] = 0.f;
          ccur[ui][rt][r] = p.c[(rowb(rt, r) + toff0) * 2 * H + (long)dir * H + uc];
        }
    }
  }
  // PF = 3: two register slots - the loads of unit tile ui + 1 are issued before tile ui is computed, so a step costs ONE exposed round trip
  // to HBM plus what the compute of a tile does not cover, instead of MAXUT dependent ones (band path, 4 tiles per wave: the input loads are
  // 1.2 of the launch's 3.8 ms, scripts/abl_lstm.py); the registers come out of the weight fragments in flight (URSE_BWD_KB2_PF3)
  constexpr int PU = PF == 3 ? 2 : (PF == 4 ? 1 : (PF ? MAXUT : 1));
  static_assert(PF != 4 || (STG && RT == 2 && sizeof(T) == 2 && KH == 1 && HPC), "PF 4: the 32-row staged-store bf16 form (row table in LDS)");
  V4 gpf[PU][RT][4];
  float cpf[PU][RT][4];
  T dhpf[PU][RT][4];
  V4 dgk[KH == 2 ? MAXUT : 1][RT][4];      // KH = 2: this wave's gate gradients of the step, until their half of the tile is due
  auto load_step = [&](int ui, int slot, int tt) {
    const int u = (w + NW * ui) * 16 + lc;
    const bool first_ = dir ? (tt == p.m.seq_len - 1) : (tt == 0);   // first step of the forward recurrence: c_{-1} = 0
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifdef BABL_HOT_INPUTS   // timing diagnostic (wrong results): every workgroup and step reads the same 16 * RT rows, i.e. the inputs are L2 hits
#if BABL_HOT_INPUTS >= 2   // ... its own rows of ONE step (12 MB over the chip); 3: the gradient store goes to those rows too
        const int row = (int)rowb(rt, r) + (p.m.seq_len / 2) * stride_i + 0 * tt;
#else
        const int row = rt * 16 + lr * 4 + r + 0 * tt;
#endif
#else
        const int row = (int)rowb(rt, r) + tt * stride_i;
#endif
#ifdef BABL_NO_P1LOAD
        gpf[slot][rt][r] = V4{}; cpf[slot][rt][r] = (float)row; dhpf[slot][rt][r] = T(row & 1);
#else
#ifdef BABL_NO_G
        gpf[slot][rt][r] = V4{};
#else
        gpf[slot][rt][r] = *reinterpret_cast<const V4*>(gates + ((long)row * ldg_i + (gcol_i + u * 4)));
#endif
#ifdef BABL_NO_C
        cpf[slot][rt][r] = (float)row;
#else
#ifdef BABL_HOT_INPUTS
        cpf[slot][rt][r] = first_ ? 0.f : p.c[(long)row * ldc_i + (hcol_i + u)];
#else
        cpf[slot][rt][r] = first_ ? 0.f : p.c[(long)(row + prev_i) * ldc_i + (hcol_i + u)];
#endif
#endif
#ifdef BABL_NO_DH
        dhpf[slot][rt][r] = T(row & 1);
#else
        dhpf[slot][rt][r] = dh[(long)row * ldd_i + (hcol_i + u)];
#endif
#endif
      }
  };
  auto load_all = [&](int tt) {
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui)
      if ((w + NW * ui) * 16 + lc < H) load_step(ui, PF ? ui : 0, tt);
  };
  // PF = 4 (round 6; the stamps put 35 % of a step on FOUR dependent input round trips, one per unit tile): the unit tiles of a wave go in PAIRS - the even
  // one's inputs are requested into registers as before and, in the same breath, the odd one's by LDS-DMA into 7 KB per wave behind the row table
  // ([32 rows][128 B] saved gates, [32][64 B] c_{t-1}, [32][32 B] dh; no register is spent on them): one round trip per pair.  Requested inside the cell
  // phase, consumed before the weight pass starts: nothing of this sits in front of a weight fragment in the wave's in-order load queue (what sank the
  // N-split's prefetch, csrc/lstm_nsplit.hip NS_PFI).
  char* pfb = smem + nbuf * R * pitch + R * (int)sizeof(int) + w * 7168;
  const unsigned pfb_u = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)pfb);
  auto dma_tile = [&](int ui, int tt) __attribute__((always_inline)) {
    const int utd = __builtin_amdgcn_readfirstlane(w) + NW * ui;          // (wave-uniform, and told so: what is built on it is scalar arithmetic; the caller checked utd < nut)
    // the row table is read again every step through an index the compiler cannot see through: left visible, the seven row lookups and the address
    // arithmetic built on them are hoisted out of the time loop and stay live across the weight pass - 22 spilled registers
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int* rowtab = reinterpret_cast<const int*>(smem + nbuf * R * pitch);
    const int ln = lane + opq;                                            // (the lane id too: piece indices and clamps are per-lane loop invariants)
    const int toff_ = tt * stride_i;
    const bool first_ = dir ? (tt == p.m.seq_len - 1) : (tt == 0);
    const int nu = H - utd * 16 < 16 ? H - utd * 16 : 16;                 // units of this tile that exist (the last tile of H = 392: 8): pieces past them re-read the last one
    {
      const int ch = ln & 7, chc = ch < nu / 2 ? ch : nu / 2 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int grow = rowtab[i * 8 + (ln >> 3)];
        grow = grow < 0 ? rowtab[0] : grow;
        bw_glds16(reinterpret_cast<const char*>(gates) + ((long)(grow + toff_) * ldg_i + (gcol_i + utd * 64)) * 2 + chc * 16, __builtin_amdgcn_readfirstlane(pfb_u + (unsigned)i * 1024u));
      }
    }
    {
      const int ch = ln & 3, chc = ch < nu / 4 ? ch : nu / 4 - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int grow = rowtab[i * 16 + (ln >> 2)];
        grow = grow < 0 ? rowtab[0] : grow;
        const int crow = grow + toff_ + (first_ ? 0 : prev_i);           // (first step of the forward recurrence: c_{-1} = 0, the value read is not used)
        bw_glds16(reinterpret_cast<const char*>(p.c) + ((long)crow * ldc_i + (hcol_i + utd * 16)) * 4 + chc * 16, __builtin_amdgcn_readfirstlane(pfb_u + 4096u + (unsigned)i * 1024u));
      }
    }
    {
      const int ch = ln & 1, chc = ch < nu / 8 ? ch : nu / 8 - 1;
      int grow = rowtab[ln >> 1];
      grow = grow < 0 ? rowtab[0] : grow;
      bw_glds16(reinterpret_cast<const char*>(dh) + ((long)(grow + toff_) * ldd_i + (hcol_i + utd * 16)) * 2 + chc * 16, __builtin_amdgcn_readfirstlane(pfb_u + 6144u));
    }
  };
  if constexpr (PF == 1) load_all(dir ? 0 : p.m.seq_len - 1);
  if constexpr (PF == 4) __syncthreads();                                 // (the row table is complete before the first DMA reads it)
  for (int step = 0; step < p.m.seq_len; ++step) {
    const int t = dir ? step : (p.m.seq_len - 1 - step);
    char* tile = smem + (step % nbuf) * R * pitch;
    BWST(0);
    if constexpr (PF == 2) load_all(t);                 // all unit tiles' inputs in ONE round trip, still inside the step
    if constexpr (PF == 3) {
      if (w * 16 + lc < H) load_step(0, 0, t);
    }
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        const int u = ut * 16 + lc;
        if constexpr (PF == 4) {
          // (wave-uniform part; the per-lane loads and reads stay inside the one `u < H` block below with the arithmetic that consumes them - assigned in a
          // block of their own, the input registers are live through the whole step for the lanes that skip it: 82 spills)
          if ((ui & 1) == 0) {
            if (ui + 1 < MAXUT && w + NW * (ui + 1) < nut) dma_tile(ui + 1, t);      // the odd partner's inputs travel with this tile's
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this tile's DMAs (the compiler does not count them); nothing younger is in flight
          }
        }
        if (u < H) {
          if constexpr (!PF) load_step(ui, 0, t);
          if constexpr (PF == 4) {
            if ((ui & 1) == 0) {
              load_step(ui, 0, t);
            } else {
              const bool first_ = dir ? (t == p.m.seq_len - 1) : (t == 0);
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int lrow = rt * 16 + lr * 4 + r;
                  gpf[0][rt][r] = *reinterpret_cast<const V4*>(pfb + lrow * 128 + lc * 8);
                  cpf[0][rt][r] = first_ ? 0.f : *reinterpret_cast<const float*>(pfb + 4096 + lrow * 64 + lc * 4);
                  dhpf[0][rt][r] = *reinterpret_cast<const T*>(pfb + 6144 + lrow * 32 + lc * 2);
                }
            }
          }
          if constexpr (PF == 3) {
            if (ui + 1 < MAXUT && (w + NW * (ui + 1)) * 16 + lc < H) load_step(ui + 1, (ui + 1) & 1, t);
          }
          const int SL = PF == 3 ? (ui & 1) : ((PF && PF != 4) ? ui : 0);         // (a constant after unrolling)
          V4 (&gpre)[RT][4] = gpf[SL];
          float (&cpre)[RT][4] = cpf[SL];
          T (&dhpre)[RT][4] = dhpf[SL];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float a[4];
              Vec4<T>::unpack(gpre[rt][r], a);
              const float iv = a[0], fv = a[1], gv = a[2], ov = a[3];
              const float dht = to_f32<T>(dhpre[rt][r]) + dhr[ui][rt][r];
              const float tc = tanhf_(ccur[ui][rt][r]);
              const float dct = dcs[ui][rt][r] + dht * ov * (1.f - tc * tc);
              float dg[4];
              dg[0] = dct * gv * iv * (1.f - iv);
              dg[1] = dct * cpre[rt][r] * fv * (1.f - fv);
              dg[2] = dct * iv * (1.f - gv * gv);
              dg[3] = dht * tc * ov * (1.f - ov);
              dcs[ui][rt][r] = dct * fv;
              ccur[ui][rt][r] = cpre[rt][r];          // c_{t-1} is the next processed step's c_t
              const V4 pk = Vec4<T>::pack(dg);
              if constexpr (KH == 2) dgk[ui][rt][r] = pk;
              else *reinterpret_cast<V4*>(tile + (rt * 16 + lr * 4 + r) * pitch + (u * 4) * ES) = pk;
#ifndef BABL_NO_STORE
#if defined(BABL_HOT_INPUTS) && BABL_HOT_INPUTS == 3
              if (rowbase[rt][r] >= 0)
                *reinterpret_cast<V4*>(gates + ((long)(rowbase[rt][r] + (p.m.seq_len / 2) * stride_i) * ldg_i + (gcol_i + u * 4))) = pk;
#else
              if (!STG && rowbase[rt][r] >= 0)
                *reinterpret_cast<V4*>(gates + ((long)(rowbase[rt][r] + t * stride_i) * ldg_i + (gcol_i + u * 4))) = pk;
#endif
#endif
            }
        }
      }
      BWST(1 + ui);
    }
    if constexpr (!STG) {
      if (step + 1 == p.m.seq_len) break;
    }
    if constexpr (KH == 2) {
      f32x4_t accs[MAXUT][RT];
#pragma unroll
      for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) accs[ui][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const int hu = H / 2, hs = nslab / 2;             // units / k slabs per half (H % 8 == 0: a slab never straddles the halves)
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ui = 0; ui < MAXUT; ++ui) {
          const int u = (w + NW * ui) * 16 + lc;
          if (u < H && (u >= hu) == (half == 1)) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *reinterpret_cast<V4*>(tile + (rt * 16 + lr * 4 + r) * pitch + ((u - half * hu) * 4) * ES) = dgk[ui][rt][r];
          }
        }
        __syncthreads();
#pragma unroll
        for (int ui = 0; ui < MAXUT; ++ui) {
          const int ut = w + NW * ui;
          if (ut < nut) {
            const char* wr = whhT + ((long)ut * nslab + half * hs) * 1024;
            const char* ar = tile + lc * pitch + 16 * lr;
            constexpr int KB = 8;
#pragma unroll 1
            for (int k0 = 0; k0 < hs; k0 += KB) {
              uint4 b[KB];
#pragma unroll
              for (int i = 0; i < KB; ++i) b[i] = *reinterpret_cast<const uint4*>(wr + ((k0 + i < hs) ? k0 + i : hs - 1) * 1024);
#pragma unroll
              for (int i = 0; i < KB; ++i) {
                if (k0 + i < hs) {
                  uint4 a[RT];
#pragma unroll
                  for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + (k0 + i) * 64);
                  mma_slab<T, RT>(a, b[i], accs[ui]);
                }
              }
            }
          }
        }
        __syncthreads();                                 // the tile is free for the other half / the next step
      }
#pragma unroll
      for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhr[ui][rt][r] = accs[ui][rt][r];
      continue;
    }
    if constexpr (PF == 1) load_all(dir ? t + 1 : t - 1);     // next step's inputs: in flight under the weight pass
    BWST(5);
    __syncthreads();
    BWST(6);
    if constexpr (STG) {
#ifndef BABL_NO_STORE
      constexpr int CPR = 4 * HPC * ES / 16;                    // 16-byte pieces of a row's direction segment
      const int* rowtab = reinterpret_cast<const int*>(smem + nbuf * R * pitch);
      for (int idx = tid; idx < R * CPR; idx += NW * 64) {
        const int row = idx / CPR, cc = idx - row * CPR;
        const int grow = rowtab[row];
        if (grow >= 0)
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(gates) + ((long)(grow + t * stride_i) * ldg_i + gcol_i) * ES + cc * 16) =
              *reinterpret_cast<const uint4*>(tile + row * pitch + cc * 16);
      }
#endif
      BWST(7);
      if (step + 1 == p.m.seq_len) break;
    }
#ifdef BABL_NO_MM
    if (p.m.seq_len > 0) continue;
#endif
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        f32x4_t acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* wr = whhT + ((long)ut * nslab) * 1024;
        const char* ar = tile + lc * pitch + 16 * lr;
        // the weight stream (L2 -> registers) is the bottleneck: keep KB fragment loads in flight per wave
#ifndef URSE_BWD_KB
#define URSE_BWD_KB 13
#endif
#ifndef URSE_BWD_KB2
#define URSE_BWD_KB2 17   // the 32-sequence geometry (8 waves, 256-VGPR budget): 49 slabs = 17 + 17 + 15, 230 VGPRs; band-path BPTT
                          // 27.15 -> 26.4 ms per step, step -1.1 ms (same-box A/B, profiles/r02_ab_bptt_kb_v1.log)
#endif
#ifndef URSE_BWD_KB8
#define URSE_BWD_KB8 17   // 16 sequences on 8 waves (256-VGPR budget, room for the prefetched inputs of 4 unit tiles): 49 = 17 + 17 + 15
#endif
#ifndef URSE_BWD_KBPF16
#define URSE_BWD_KBPF16 8   // 16 / 13 waves with prefetched inputs: what the 128-VGPR cap leaves for fragments in flight
#endif
#ifndef URSE_BWD_KBPF12
#define URSE_BWD_KBPF12 14
#endif
#ifndef URSE_BWD_KB2_PF3
#define URSE_BWD_KB2_PF3 9
#endif
#ifndef URSE_BWD_KB2_STG
#define URSE_BWD_KB2_STG 25   // the staged-store form of the 32-sequence geometry has the registers for 49 = 25 + 24 fragments in flight (246 VGPRs, no
                              // spill): 3.69 -> 3.61 ms per launch, same bits (scripts/diag/kb2_sweep.sh; 19 and 21 leave ragged last batches and lose)
#endif
#ifndef URSE_BWD_KB2_PF4
#define URSE_BWD_KB2_PF4 17   // the paired-tile form: what its DMA addresses and LDS pointer leave for fragments in flight (49 = 17 + 17 + 15)
#endif
#ifndef URSE_BWD_KB3
#define URSE_BWD_KB3 13   // 48 rows on four waves (512 registers per wave): what the per-row state of 7 x 3 tiles leaves for fragments in flight
#endif
        constexpr int KB = (sizeof(T) == 2) ? (RT >= 3 ? URSE_BWD_KB3 : RT >= 2 ? (PF == 3 ? URSE_BWD_KB2_PF3 : (PF == 4 ? URSE_BWD_KB2_PF4 : (STG ? URSE_BWD_KB2_STG : URSE_BWD_KB2))) : (NW == 8 ? URSE_BWD_KB8 : (PF == 1 ? (NW == 12 ? URSE_BWD_KBPF12 : URSE_BWD_KBPF16) : URSE_BWD_KB))) : 8;
        #pragma unroll 1
        for (int k0 = 0; k0 < nslab; k0 += KB) {
          uint4 b[KB];
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (k0 + i < nslab) ? k0 + i : nslab - 1;
#ifdef BABL_PIN_W
            b[i] = *reinterpret_cast<const uint4*>(whhT + (ks & 7) * 1024);   // diagnostic: 8 KB working set
#else
            b[i] = *reinterpret_cast<const uint4*>(wr + ks * 1024);
#endif
          }
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            if (k0 + i < nslab) {
              uint4 a[RT];
#pragma unroll
              for (int rt = 0; rt < RT; ++rt)
                a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + (k0 + i) * 64);
              mma_slab<T, RT>(a, b[i], acc);
            }
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhr[ui][rt][r] = acc[rt][r];
      }
      BWST(8 + ui);
    }
    if (nbuf == 1) __syncthreads();
  }
}

// ---- BPTT with the waves of a workgroup in two ROLES (bf16, H = 392, 32 sequences per workgroup; round 6) ---------------------------------
// lstm_bwd_kernel uses two resources one after the other inside a workgroup: the cell phase waits for HBM (the step's saved gates, c_{t-1} and dh:
// 179 KB per step and workgroup; over the launch 6.0 GB read + 2.75 GB written = half its time at 5 TB/s) with the L2-bound weight pass idle, then the
// weight pass (1.57 MB of W_hh^T per step through the CU's memory path) runs with HBM idle - stamps: 35 % / 59 % of a step, profiles/
// r06_stamps_tn224_bwdband_rwx_v1.log.  The inputs do not depend on the recurrence, only the arithmetic does; but a wave's loads return in ORDER, so a wave
// that prefetches them makes its own weight fragments (L2 hits) wait for HBM (the N-split's prefetch, the paired tiles: both measured slower).  So the two
// streams get their own waves:
//   seven CELL waves own the per-element state (unit w * 56 .. + 55 of wave w, 28 elements per lane: dc and c in registers) and everything that touches HBM:
//     after the step's barrier they store the gate gradients (16-byte row pieces out of the LDS tile) and request the NEXT step's inputs into registers -
//     both travel while the product runs; at the top of a step they read dh_rec from LDS, do the cell arithmetic and write the tile;
//   one PRODUCT wave streams W_hh^T (25 unit tiles x 49 fragments, a whole tile's 49 in flight, refilled slot by slot across tiles and steps) against the
//     tile and writes dh_rec [32][392] f32 to LDS.
// Two barriers per step (tile complete / dh_rec complete).  Arithmetic and summation order are lstm_bwd_kernel's: bit-identical gate gradients.
template <int H>
__global__ void __launch_bounds__(512) lstm_bwd_roles_kernel(LstmBwdArgs p) {
  constexpr int R = 32, NUT = (H + 15) / 16, G4 = 4 * H, NSLAB = G4 * 2 / 64, PITCH = lds_frag_pitch(G4 * 2);
  constexpr int NCW = 7, UPW = H / NCW, EPL = R * UPW / 64, KC = UPW / 8;        // cell waves, units per cell wave (56), elements per lane (28), 8-unit groups per wave (7)
  constexpr int DP = H + 12;                                                    // dh_rec row pitch in floats (404: the product wave's 4-row groups fall on different banks)
  static_assert(H % NCW == 0 && UPW % 8 == 0 && EPL * 64 == R * UPW && EPL == 4 * KC, "element map: row = (k / KC) * 8 + lane / 8, unit = (k % KC) * 8 + lane % 8");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                            // [32][PITCH] bf16 gate gradients of the step (MFMA A operand)
  float* dhrec = reinterpret_cast<float*>(smem + R * PITCH);                   // [32][DP] f32
  int* rowtab = reinterpret_cast<int*>(smem + R * PITCH + R * DP * 4);         // [32] row of (sequence, t = 0), -1 beyond n_seq
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int dir, tile_;
  xcd_dir_tile(p.xcd, dir, tile_);
  const int s0 = tile_ * R;
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.m.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  if (tid < R) {
    const int seq = s0 + tid;
    rowtab[tid] = seq < p.m.n_seq ? (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner)) : -1;
  }
  for (int i = tid; i < R * DP; i += 512) dhrec[i] = 0.f;                       // dh_rec of the first step
  for (int i = tid; i < R * PITCH / 16; i += 512) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);      // (the K padding past 4H stays zero)
  __syncthreads();
  const bf16_t* dh = reinterpret_cast<const bf16_t*>(p.dh);
  bf16_t* gates = reinterpret_cast<bf16_t*>(p.gates);

  if (w < NCW) {
    // ================= cell waves =================
    const int r8 = lane >> 3, c8 = lane & 7;
    int rowb[4];                                                                // row of this lane's sequences (kr = 0 .. 3: sequence kr * 8 + r8) at t = 0, clamped beyond n_seq
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      int seq = s0 + kr * 8 + r8;
      if (seq >= p.m.n_seq) seq = p.m.n_seq - 1;
      rowb[kr] = (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner));
    }
    const int u0 = w * UPW + c8;                                                // unit of element k: u0 + (k % KC) * 8
    float dcs[EPL], ccur[EPL];
    uint2 gp[EPL];
    float cp[EPL];
    bf16_t dp[EPL];
    // buffer descriptors over the three input matrices (32-bit byte offsets: checked on the host): a lane's 28 elements are FOUR row bases + an
    // instruction's immediate (element k: row base k / KC, 8-unit group k % KC) - twelve address registers per step instead of 84 64-bit pointers
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)0xFFFFF000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, (int)0xFFFFF000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)0xFFFFF000u, 0x00020000);
    auto load_inputs = [&](int tt) __attribute__((always_inline)) {
      const int toff_ = tt * stride_i;
      const bool first_ = dir ? (tt == p.m.seq_len - 1) : (tt == 0);          // first step of the forward recurrence: c_{-1} = 0
      unsigned og[4], oc[4], od[4];
#pragma unroll
      for (int kr = 0; kr < 4; ++kr) {
        const int row = rowb[kr] + toff_;
        og[kr] = (unsigned)(row * ldg_i + gcol_i + u0 * 4) * 2u;
        oc[kr] = (unsigned)((row + (first_ ? 0 : prev_i)) * ldc_i + hcol_i + u0) * 4u;
        od[kr] = (unsigned)(row * ldd_i + hcol_i + u0) * 2u;
      }
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ g2 = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)(og[k / KC] + (unsigned)((k % KC) * 64)), 0, 0);
        gp[k] = make_uint2(g2[0], g2[1]);
        const float cv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(oc[k / KC] + (unsigned)((k % KC) * 32)), 0, 0));
        cp[k] = first_ ? 0.f : cv;
        dp[k] = (bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_d, (int)(od[k / KC] + (unsigned)((k % KC) * 16)), 0, 0);
      }
    };
    {
      const int t0 = dir ? 0 : p.m.seq_len - 1;
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        dcs[k] = 0.f;
        ccur[k] = p.c[(long)(rowb[k / KC] + t0 * stride_i) * ldc_i + (hcol_i + u0 + (k % KC) * 8)];
      }
      load_inputs(t0);
    }
    constexpr int CPR = G4 * 2 / 16, CTHR = NCW * 64;                           // 16-byte pieces of a row's direction segment (196), cell threads (448)
    for (int step = 0; step < p.m.seq_len; ++step) {
      const int t = dir ? step : (p.m.seq_len - 1 - step);
      const int toff = t * stride_i;
      // ---- cell arithmetic on inputs that were requested a product ago; dh_rec from the product wave
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int lrow = (k / KC) * 8 + r8, u = u0 + (k % KC) * 8;
        float a[4];
        Vec4<bf16_t>::unpack(gp[k], a);
        const float iv = a[0], fv = a[1], gv = a[2], ov = a[3];
        const float dht = bf16_to_f32(dp[k]) + dhrec[lrow * DP + u];
        const float tc = tanhf_(ccur[k]);
        const float dct = dcs[k] + dht * ov * (1.f - tc * tc);
        float dg[4];
        dg[0] = dct * gv * iv * (1.f - iv);
        dg[1] = dct * cp[k] * fv * (1.f - fv);
        dg[2] = dct * iv * (1.f - gv * gv);
        dg[3] = dht * tc * ov * (1.f - ov);
        dcs[k] = dct * fv;
        ccur[k] = cp[k];                                                        // c_{t-1} is the next processed step's c_t
        *reinterpret_cast<uint2*>(tile + lrow * PITCH + u * 8) = Vec4<bf16_t>::pack(dg);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                             // A: the tile is complete, dh_rec has been consumed
      // ---- the step's gate gradients leave (16-byte row pieces), the next step's inputs are requested: both under the product
      for (int idx = tid; idx < R * CPR; idx += CTHR) {
        const int row = idx / CPR, cc = idx - row * CPR;
        const int grow = rowtab[row];
        if (grow >= 0)
          *reinterpret_cast<uint4*>(reinterpret_cast<char*>(gates) + ((long)(grow + toff) * ldg_i + gcol_i) * 2 + cc * 16) =
              *reinterpret_cast<const uint4*>(tile + row * PITCH + cc * 16);
      }
      if (step + 1 == p.m.seq_len) break;
      load_inputs(dir ? t + 1 : t - 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // (the tile reads of the stores are done before the barrier lets the next step write it)
      __builtin_amdgcn_s_barrier();                                             // B: dh_rec of this step is complete
    }
    return;
  }

  // ================= product wave =================
  const int lr = lane >> 4, lc = lane & 15;
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * NUT * NSLAB) * 1024 + lane * 16;
  const char* ar = tile + lc * PITCH + 16 * lr;
  // ONE wave has to keep the CU's memory path busy (68 GB/s x ~0.7 us = 48 KB in flight): a whole unit tile's 49 fragments sit in registers, and the slot of
  // a fragment is refilled with the NEXT tile's fragment the moment its MFMAs have issued - 49 KB in flight without a gap between tiles, nor between steps
  // (the weights do not depend on the step: the first tile of step t + 1 is requested while the last tile of step t is multiplied)
  uint4 b[NSLAB];
#pragma unroll
  for (int i = 0; i < NSLAB; ++i) b[i] = *reinterpret_cast<const uint4*>(whhT + (long)i * 1024);
  for (int step = 0; step < p.m.seq_len; ++step) {
    __builtin_amdgcn_s_barrier();                                               // A
    if (step + 1 == p.m.seq_len) break;
#pragma unroll 1
    for (int ut = 0; ut < NUT; ++ut) {
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      const char* wn = whhT + (long)(ut + 1 < NUT ? ut + 1 : 0) * NSLAB * 1024;  // the tile that follows (tile 0 of the next step behind the last one)
#pragma unroll
      for (int i = 0; i < NSLAB; ++i) {
        uint4 a[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + i * 64);
        mma_slab<bf16_t, 2>(a, b[i], acc);
        b[i] = *reinterpret_cast<const uint4*>(wn + (long)i * 1024);
      }
      const int u = ut * 16 + lc;
      if (u < H) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhrec[(rt * 16 + lr * 4 + r) * DP + u] = acc[rt][r];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                               // B
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              // (the last refills are never consumed; they must not outlive the wave)
}

// ---- BPTT, transposed accumulator layout (bf16) -------------------------------------------------------------------------------
// lstm_bwd_kernel multiplies dgates (A operand, rows = sequences) with W_hh fragments (B operand, columns = units): the MFMA C
// layout then gives a lane ONE unit and FOUR sequences, and every per-(sequence, unit) input of the cell update - saved gates
// (8 bytes), c_{t-1} (4), dh (2) - is a separate load per sequence: 48 narrow load instructions per lane and step at four unit
// tiles per wave, 16 narrow stores.  Measured (profiles/r03_exp_bptt_*.log): with the recurrent product switched off the step
// still costs 5.7 us, and each of the three input arrays costs about the same whatever its width - the vector memory pipe is
// bound by wave-INSTRUCTIONS (~28 cycles each for these 4-row scatter patterns), not by bytes or lines.
// Here the operands swap roles: A = the W_hh fragment (rows = units of the tile), B = the dgates fragment (columns = sequences).
// The fragments' register contents are the same as before (both MFMA operand layouts index "row l % 16 / 8 k's by l / 16"), only
// the accumulator changes: lane (lr = l / 16, lc = l % 16) holds units 4 lr .. 4 lr + 3 of ONE sequence lc.  Four consecutive
// units are contiguous in every array, so a lane loads c_{t-1} with one 16-byte load, dh with one 8-byte load, the saved gates
// with two 16-byte loads, and stores its 16 gate gradients with two 16-byte stores: 16 + 8 instead of 48 + 16 instructions.
// The next step's inputs are prefetched in front of the weight pass (PF of lstm_bwd_kernel).  H % 4 == 0.
template <int RT, int MAXUT, int NW, int HC>
__global__ void __launch_bounds__(NW * 64) lstm_bwd_tr_kernel(LstmBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  int dir, tile_;
  xcd_dir_tile(p.xcd, dir, tile_);
  const int s0 = tile_ * R;
  const int H = HC ? HC : p.H, nut = (H + 15) >> 4, G4 = 4 * H;
  const int pitch = lds_frag_pitch(G4 * 2);
  const int nbuf = p.dbuf ? 2 : 1;
  float dcs[MAXUT][RT][4], dhr[MAXUT][RT][4], ccur[MAXUT][RT][4];
  int rowbase[RT];                         // this lane's sequence per row tile; negative: beyond n_seq (clamped, never stored)
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int seq = s0 + rt * 16 + lc;
    const bool ok = seq < p.m.n_seq;
    if (!ok) seq = p.m.n_seq - 1;
    const int rb = (int)((seq / p.m.inner) * p.m.outer + (seq % p.m.inner));
    rowbase[rt] = ok ? rb : -rb - 1;
  }
  auto rowb = [&](int rt) -> int { return rowbase[rt] >= 0 ? rowbase[rt] : -(rowbase[rt] + 1); };
  const int nslab = G4 * 2 / 64;
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * nut * nslab) * 1024 + lane * 16;
  const bf16_t* dh = reinterpret_cast<const bf16_t*>(p.dh);
  bf16_t* gates = reinterpret_cast<bf16_t*>(p.gates);
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.m.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  auto quad = [&](int ui) -> int { return (w + NW * ui) * 16 + lr * 4; };     // first of this lane's four units in tile ui

  {
    const int t0 = dir ? 0 : p.m.seq_len - 1;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int u0 = quad(ui) < H ? quad(ui) : H - 4;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 cv = *reinterpret_cast<const float4*>(p.c + (long)(rowb(rt) + t0 * stride_i) * ldc_i + (hcol_i + u0));
        const float c4[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dcs[ui][rt][r] = 0.f;
          dhr[ui][rt][r] = 0.f;
          ccur[ui][rt][r] = c4[r];
        }
      }
    }
  }
  uint4 gpf[MAXUT][RT][2];
  float4 cpf[MAXUT][RT];
  uint2 dpf[MAXUT][RT];
  auto load_all = [&](int tt) {
    const bool first_ = dir ? (tt == p.m.seq_len - 1) : (tt == 0);   // first step of the forward recurrence: c_{-1} = 0
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int u0 = quad(ui);
      if (u0 < H) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = rowb(rt) + tt * stride_i;
          const bf16_t* gp = gates + ((long)row * ldg_i + (gcol_i + u0 * 4));
          gpf[ui][rt][0] = *reinterpret_cast<const uint4*>(gp);
          gpf[ui][rt][1] = *reinterpret_cast<const uint4*>(gp + 8);
          cpf[ui][rt] = first_ ? make_float4(0.f, 0.f, 0.f, 0.f)
                               : *reinterpret_cast<const float4*>(p.c + (long)(row + prev_i) * ldc_i + (hcol_i + u0));
          dpf[ui][rt] = *reinterpret_cast<const uint2*>(dh + (long)row * ldd_i + (hcol_i + u0));
        }
      }
    }
  };
  load_all(dir ? 0 : p.m.seq_len - 1);
  for (int step = 0; step < p.m.seq_len; ++step) {
    const int t = dir ? step : (p.m.seq_len - 1 - step);
    char* tile = smem + (step % nbuf) * R * pitch;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int u0 = quad(ui);
      if (u0 < H) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const unsigned gw[8] = {gpf[ui][rt][0].x, gpf[ui][rt][0].y, gpf[ui][rt][0].z, gpf[ui][rt][0].w,
                                  gpf[ui][rt][1].x, gpf[ui][rt][1].y, gpf[ui][rt][1].z, gpf[ui][rt][1].w};
          const float c4[4] = {cpf[ui][rt].x, cpf[ui][rt].y, cpf[ui][rt].z, cpf[ui][rt].w};
          const unsigned dw[2] = {dpf[ui][rt].x, dpf[ui][rt].y};
          unsigned ow[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float iv = __uint_as_float(gw[2 * r] << 16), fv = __uint_as_float(gw[2 * r] & 0xffff0000u);
            const float gv = __uint_as_float(gw[2 * r + 1] << 16), ov = __uint_as_float(gw[2 * r + 1] & 0xffff0000u);
            const float dhv = __uint_as_float((r & 1) ? (dw[r >> 1] & 0xffff0000u) : (dw[r >> 1] << 16));
            const float dht = dhv + dhr[ui][rt][r];
            const float tc = tanhf_(ccur[ui][rt][r]);
            const float dct = dcs[ui][rt][r] + dht * ov * (1.f - tc * tc);
            const float d0 = dct * gv * iv * (1.f - iv);
            const float d1 = dct * c4[r] * fv * (1.f - fv);
            const float d2 = dct * iv * (1.f - gv * gv);
            const float d3 = dht * tc * ov * (1.f - ov);
            dcs[ui][rt][r] = dct * fv;
            ccur[ui][rt][r] = c4[r];                 // c_{t-1} is the next processed step's c_t
            ow[2 * r] = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
            ow[2 * r + 1] = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
          }
          const uint4 o0 = make_uint4(ow[0], ow[1], ow[2], ow[3]), o1 = make_uint4(ow[4], ow[5], ow[6], ow[7]);
          char* tp = tile + (rt * 16 + lc) * pitch + (u0 * 4) * 2;
          *reinterpret_cast<uint4*>(tp) = o0;
          *reinterpret_cast<uint4*>(tp + 16) = o1;
#ifndef BABL_NO_STORE
          if (rowbase[rt] >= 0) {
            bf16_t* gp = gates + ((long)(rowbase[rt] + t * stride_i) * ldg_i + (gcol_i + u0 * 4));
            *reinterpret_cast<uint4*>(gp) = o0;
            *reinterpret_cast<uint4*>(gp + 8) = o1;
          }
#endif
        }
      }
    }
    if (step + 1 == p.m.seq_len) break;
    load_all(dir ? t + 1 : t - 1);               // next step's inputs: older than every weight fragment of this step's pass
    __syncthreads();
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        f32x4_t acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* wr = whhT + ((long)ut * nslab) * 1024;
        const char* ar = tile + lc * pitch + 16 * lr;
#ifndef URSE_BWD_TR_KB
#define URSE_BWD_TR_KB 17
#endif
        constexpr int KB = URSE_BWD_TR_KB;
        #pragma unroll 1
        for (int k0 = 0; k0 < nslab; k0 += KB) {
          uint4 b[KB];
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int ks = (k0 + i < nslab) ? k0 + i : nslab - 1;
            b[i] = *reinterpret_cast<const uint4*>(wr + ks * 1024);
          }
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            if (k0 + i < nslab) {
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) {
                const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + (k0 + i) * 64);
                // A = weight fragment (rows = units), B = dgates fragment (columns = sequences): C[unit 4 lr + r][sequence lc]
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[i]),
                                                                  __builtin_bit_cast(bf16x8_t, a), acc[rt], 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhr[ui][rt][r] = acc[rt][r];
      }
    }
    if (nbuf == 1) __syncthreads();
  }
}

// Packs one bidirectional LSTM's f32 master weights into the layouts the kernels read.
//  out 0: wih_p  [8H][Np]   rows permuted to (dir, unit, gate), K zero padded
//  out 1: wihT_p [N][8H]    same permutation on the columns
//  out 2: bias   [8H] f32   b_ih + b_hh, permuted
//  out 3: whh fragments     [2][nut][nslab][4][64 lanes][16 B]   (B operand of h * W_hh^T)
//  out 4: whhT fragments    [2][nut][nslabT][64 lanes][16 B]     (B operand of dgates * W_hh, k = unit*4+gate)
// T: format of the forward layouts (wih_p, whh_f); TB: of the backward layouts (wihT_p, whhT_f) - the same, except bf16 under f16 forward operands
template <typename T, typename TB = T>
__device__ __forceinline__ void lstm_pack_dev(const float* __restrict__ wih, const float* __restrict__ whh,
                                              const float* __restrict__ bih, const float* __restrict__ bhh,
                                              T* __restrict__ wih_p, TB* __restrict__ wihT_p,
                                              float* __restrict__ bias, T* __restrict__ whh_f,
                                              TB* __restrict__ whhT_f, int N, int Np, int H, int Hp) {
  static_assert(sizeof(T) == sizeof(TB), "one element size per pack");
  constexpr int ES = sizeof(T), EPL = 16 / ES, SK = 64 / ES;
  const int nut = (H + 15) >> 4, G4 = 4 * H;
  const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int which = blockIdx.y;
  if (which == 0) {
    for (long idx = i0; idx < (long)2 * G4 * Np; idx += stride) {
      const int rp = (int)(idx / Np), n = (int)(idx - (long)rp * Np);
      const int d = rp / G4, r = rp - d * G4, u = r >> 2, g = r & 3;
      wih_p[idx] = from_f32<T>(n < N ? wih[((long)d * G4 + g * H + u) * N + n] : 0.f);
    }
  } else if (which == 1) {
    for (long idx = i0; idx < (long)N * 2 * G4; idx += stride) {
      const int n = (int)(idx / (2 * G4)), rp = (int)(idx - (long)n * 2 * G4);
      const int d = rp / G4, r = rp - d * G4, u = r >> 2, g = r & 3;
      wihT_p[idx] = from_f32<TB>(wih[((long)d * G4 + g * H + u) * N + n]);
    }
  } else if (which == 2) {
    for (long idx = i0; idx < 2 * G4; idx += stride) {
      const int d = (int)idx / G4, r = (int)idx - d * G4, u = r >> 2, g = r & 3;
      const int src = d * G4 + g * H + u;
      bias[idx] = bih[src] + bhh[src];
    }
  } else if (which == 3) {
    const int nslab = Hp / SK;
    const long total = (long)2 * nut * nslab * 4 * 64 * EPL;
    for (long idx = i0; idx < total; idx += stride) {
      long q = idx;
      const int j = (int)(q % EPL); q /= EPL;
      const int lane = (int)(q % 64); q /= 64;
      const int g = (int)(q % 4); q /= 4;
      const int ks = (int)(q % nslab); q /= nslab;
      const int ut = (int)(q % nut);
      const int d = (int)(q / nut);
      const int u = ut * 16 + (lane & 15), k = ks * SK + EPL * (lane >> 4) + j;
      whh_f[idx] = from_f32<T>((u < H && k < H) ? whh[((long)d * G4 + g * H + u) * H + k] : 0.f);
    }
  } else {
    const int nslab = G4 / SK;
    const long total = (long)2 * nut * nslab * 64 * EPL;
    for (long idx = i0; idx < total; idx += stride) {
      long q = idx;
      const int j = (int)(q % EPL); q /= EPL;
      const int lane = (int)(q % 64); q /= 64;
      const int ks = (int)(q % nslab); q /= nslab;
      const int ut = (int)(q % nut);
      const int d = (int)(q / nut);
      const int n = ut * 16 + (lane & 15), kk = ks * SK + EPL * (lane >> 4) + j;
      const int up = kk >> 2, gp = kk & 3;
      whhT_f[idx] = from_f32<TB>(n < H ? whh[((long)d * G4 + gp * H + up) * H + n] : 0.f);
    }
  }
}

template <typename T, typename TB = T>
__global__ void __launch_bounds__(256) lstm_pack_kernel(const float* __restrict__ wih, const float* __restrict__ whh,
                                                        const float* __restrict__ bih, const float* __restrict__ bhh,
                                                        T* __restrict__ wih_p, TB* __restrict__ wihT_p,
                                                        float* __restrict__ bias, T* __restrict__ whh_f,
                                                        TB* __restrict__ whhT_f, int N, int Np, int H, int Hp) {
  lstm_pack_dev<T, TB>(wih, whh, bih, bhh, wih_p, wihT_p, bias, whh_f, whhT_f, N, Np, H, Hp);
}
// all LSTMs of a model in one launch: blockIdx.z = row of the pointer table (the model re-packs its 12 LSTMs after every optimizer step)
template <typename T, typename TB = T>
__global__ void __launch_bounds__(256) lstm_pack_multi_kernel(const PackRow* __restrict__ tab, int N, int Np, int H, int Hp) {
  const PackRow r = tab[blockIdx.z];
  lstm_pack_dev<T, TB>(r.wih, r.whh, r.bih, r.bhh, (T*)r.wih_p, (TB*)r.wihT_p, r.bias, (T*)r.whh_f, (TB*)r.whhT_f, N, Np, H, Hp);
}

template <typename K>
static void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}

template <typename T, int RT, int NW>
static int launch_fwd(const LstmFwdArgs& p, hipStream_t st) {
  constexpr int R = 16 * RT;
  const size_t lds = (size_t)2 * R * lds_frag_pitch(p.Hp * (int)sizeof(T));
  URSE_CHECK_ARG(lds <= 160 * 1024, "urse_lstm_fwd: Hp %d with %d rows exceeds LDS", p.Hp, R);
  dim3 grid(ceil_div(p.m.n_seq, R), 2);
  const int upw = ((p.H + 15) / 16 + NW - 1) / NW;   // unit tiles per wave
#define URSE_LF(MU)                                                                              \
  {                                                                                              \
    static bool once = (allow_big_lds(lstm_fwd_kernel<T, RT, MU, NW>), true);                    \
    (void)once;                                                                                  \
    hipLaunchKernelGGL((lstm_fwd_kernel<T, RT, MU, NW>), grid, dim3(NW * 64), lds, st, p);       \
  }
  if (upw <= 2) URSE_LF(2) else if (upw <= 4) URSE_LF(4) else URSE_LF(6)
#undef URSE_LF
  URSE_CHECK_LAUNCH("urse_lstm_fwd");
  return URSE_OK;
}

// URSE_BWD_VARIANT (experiments): 0 = default dispatch, 1 = 16 waves + prefetch, 2 = 8 waves + prefetch on the time path
// Round 6: the variants (16 waves + prefetch, 13 x 2 / 12 x 3 unit tiles, in-step single round trip, transposed accumulators, two-slot input loads, 48 rows on four
// 512-register waves) were each measured slower than the default dispatch (docs/HISTORY.md) and are instantiated only in variant builds
// (-DURSE_EXPERIMENTS through scripts/build_variant.sh): the shipped library holds the kernels the dispatch below can reach without a switch.
#ifdef URSE_EXPERIMENTS
static int bwd_variant() {
  const char* e = getenv("URSE_BWD_VARIANT");
  return e ? atoi(e) : 0;
}
#else
static constexpr int bwd_variant() { return 0; }
#endif

template <typename T, int RT, int NW>
static int launch_bwd(const LstmBwdArgs& p, hipStream_t st) {
  constexpr int R = 16 * RT;
  size_t lds = (size_t)R * lds_frag_pitch(4 * p.H * (int)sizeof(T));
  if constexpr (sizeof(T) == 4 && RT == 1 && NW == 16) {
    if (lds > 160 * 1024 && p.H % 8 == 0 && (size_t)R * lds_frag_pitch(2 * p.H * 4) <= 160 * 1024) {
      // exact-f32 mode at the flow model's H = 768: half the gate-gradient columns in LDS at a time (KH = 2)
      LstmBwdArgs pa = p;
      pa.dbuf = 0;
      pa.xcd = xcd_dir_env() & 1;
      const size_t lds2 = (size_t)R * lds_frag_pitch(2 * p.H * 4);
      dim3 grid(ceil_div(p.m.n_seq, R), 2);
      const int upw = ((p.H + 15) / 16 + NW - 1) / NW;
      if (upw <= 3) {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 3, 16, 0, 0, 0, 0, 2>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 3, 16, 0, 0, 0, 0, 2>), grid, dim3(NW * 64), lds2, st, pa);
      } else {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 6, 16, 0, 0, 0, 0, 2>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 6, 16, 0, 0, 0, 0, 2>), grid, dim3(NW * 64), lds2, st, pa);
      }
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
  URSE_CHECK_ARG(lds <= 160 * 1024, "urse_lstm_bwd: H %d with %d rows exceeds LDS", p.H, R);
  LstmBwdArgs pa = p;
  pa.dbuf = (2 * lds <= 150 * 1024) ? 1 : 0;
  pa.xcd = xcd_dir_env() & 1;
  if (pa.dbuf) lds *= 2;
  dim3 grid(ceil_div(p.m.n_seq, R), 2);
  const int upw = ((p.H + 15) / 16 + NW - 1) / NW;
#define URSE_LB(MU)                                                                              \
  {                                                                                              \
    static bool once = (allow_big_lds(lstm_bwd_kernel<T, RT, MU, NW>), true);                    \
    (void)once;                                                                                  \
    hipLaunchKernelGGL((lstm_bwd_kernel<T, RT, MU, NW>), grid, dim3(NW * 64), lds, st, pa);      \
  }
  if constexpr (sizeof(T) == 2 && NW == 8 && RT == 2) {
    if (p.H == 392 && upw <= 4) {
      static const bool stg_on = !(getenv("URSE_BWD_STAGED_STORES") && atoi(getenv("URSE_BWD_STAGED_STORES")) == 0);     // (A/B switch)
      const bool stg = stg_on && (p.ldg * 2) % 16 == 0 && (reinterpret_cast<uintptr_t>(p.gates) % 16) == 0;
#ifdef URSE_EXPERIMENTS
      if (stg && bwd_variant() == 7) {        // experiment: two-slot pipelined input loads (PF = 3)
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 3, 1>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 3, 1>), grid, dim3(NW * 64), lds + R * sizeof(int), st, pa);
      } else
#endif
#ifdef URSE_EXPERIMENTS
      // round 6, measured and dropped (variant builds only; URSE_BWD_ROLES=1): lstm_bwd_roles_kernel - parity-green (1 bf16 ulp from this kernel: other fma contractions), no
      // spill, and TWICE as slow: 8.15 against 4.08 ms per launch (profiles/r06_ab_bwd_roles_v1.log).  Its one product wave keeps a whole tile's 49 fragments in flight and
      // refills them across tiles and steps - and that is all ONE wave's load queue holds (vmcnt counts to 63): 49 KB against the ~200 KB the eight waves of this kernel keep
      // in flight to reach the CU's 68 GB/s.  The weight stream needs the registers of many waves, the per-element state needs them too, and 32 rows leave no LDS for either.
      const bool roles = getenv("URSE_BWD_ROLES") && atoi(getenv("URSE_BWD_ROLES")) != 0;
      // (its cell waves address the three input matrices with 32-bit byte offsets)
      const long rows_ = p.m.stride * (p.m.seq_len - 1) + ((long)(p.m.n_seq - 1) / p.m.inner) * p.m.outer + (p.m.n_seq - 1) % p.m.inner + 1;
      if (stg && roles && (size_t)R * lds_frag_pitch(4 * 392 * 2) + R * (392 + 12) * 4 + R * sizeof(int) <= 160 * 1024 &&
          rows_ * p.ldg * 2 < 0xFFFFF000L && rows_ * 2L * p.H * 4 < 0xFFFFF000L && rows_ * p.ldd * 2 < 0xFFFFF000L && rows_ * p.ldg < (1L << 31)) {
        static bool once = (allow_big_lds(lstm_bwd_roles_kernel<392>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_roles_kernel<392>), grid, dim3(512), (size_t)R * lds_frag_pitch(4 * 392 * 2) + R * (392 + 12) * 4 + R * sizeof(int), st, pa);
        URSE_CHECK_LAUNCH("urse_lstm_bwd");
        return URSE_OK;
      }
      // round 6, measured and dropped (variant builds only): unit tiles in PAIRS, the odd one's inputs by LDS-DMA beside the even one's register loads (PF = 4; parity-green,
      // no spill at 17 fragments in flight).  3.65 vs 3.48 ms alone, 25.8 vs 24.4 ms of band BPTT per step (profiles/r06_ab_bwd_pairs_v1.log): the stamps of the paired form
      // show the first tile's wait growing from 8.1 k to 12.4 k cycles - what looked like four dependent LATENCIES is the throughput of small-sector HBM gathers (8 + 4 + 2
      // bytes per row and unit): asking for two tiles at once takes as long as asking twice.
      static const bool pairs = getenv("URSE_BWD_PAIRS") && atoi(getenv("URSE_BWD_PAIRS")) != 0;
      if (stg && pairs && !pa.dbuf && lds + R * sizeof(int) + NW * 7168 <= 160 * 1024 && (p.ldd * 2) % 16 == 0 && (reinterpret_cast<uintptr_t>(p.dh) % 16) == 0 &&
          (reinterpret_cast<uintptr_t>(p.c) % 16) == 0) {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 4, 1>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 4, 1>), grid, dim3(NW * 64), lds + R * sizeof(int) + NW * 7168, st, pa);
      } else
#endif
      if (stg) {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 0, 1>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 2, 4, 8, 0, 392, 0, 1>), grid, dim3(NW * 64), lds + R * sizeof(int), st, pa);
      } else {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 2, 4, 8, 0, 392>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 2, 4, 8, 0, 392>), grid, dim3(NW * 64), lds, st, pa);
      }
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
#ifdef URSE_EXPERIMENTS
  if constexpr (sizeof(T) == 2 && NW == 4 && RT == 3) {
    // 48 sequences per workgroup on FOUR waves (one per SIMD: the 512-register budget holds 7 unit tiles x 3 row tiles of per-row state,
    // which spilled at the 256 of two waves per SIMD, profiles/r04_exp_bwd_band_rows48_v1.log): 1.5x the rows per pass over W_hh^T
    static bool once = (allow_big_lds(lstm_bwd_kernel<T, 3, 7, 4, 0, 392, 0, 1>), true);
    (void)once;
    pa.dbuf = 0;
    hipLaunchKernelGGL((lstm_bwd_kernel<T, 3, 7, 4, 0, 392, 0, 1>), grid, dim3(NW * 64), (size_t)R * lds_frag_pitch(4 * p.H * 2) + R * sizeof(int), st, pa);
    URSE_CHECK_LAUNCH("urse_lstm_bwd");
    return URSE_OK;
  } else
#endif
  {
  if constexpr (sizeof(T) == 2 && NW == 8 && RT == 1) {
#ifdef URSE_EXPERIMENTS
    if (p.H == 392 && bwd_variant() == 6) {     // transposed accumulator layout: wide loads / stores
      static bool once = (allow_big_lds(lstm_bwd_tr_kernel<1, 4, 8, 392>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_tr_kernel<1, 4, 8, 392>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
#endif
    if (p.H == 392) {     // 16 sequences on 8 waves with the next step's inputs prefetched (PF = 1)
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 4, 8, 392, 0, 1>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 4, 8, 392, 0, 1>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
  if constexpr (sizeof(T) == 2 && NW == 16 && RT == 1) {
#ifdef URSE_EXPERIMENTS
    if (p.H == 392 && (bwd_variant() == 4 || bwd_variant() == 5)) {     // experiments: 13 x 2 / 12 x 3 unit tiles, prefetched inputs
      const int nw = bwd_variant() == 4 ? 13 : 12;
      if (nw == 13) {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 2, 13, 392, 0, 1>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 2, 13, 392, 0, 1>), grid, dim3(nw * 64), lds, st, pa);
      } else {
        static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 3, 12, 392, 0, 1>), true);
        (void)once;
        hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 3, 12, 392, 0, 1>), grid, dim3(nw * 64), lds, st, pa);
      }
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
    if (p.H == 392 && bwd_variant() == 3) {     // experiment: both unit tiles' inputs in one round trip at the top of the step
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 2, 16, 392, 0, 2>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 2, 16, 392, 0, 2>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
    if (p.H == 392 && bwd_variant() == 1) {     // experiment: prefetch at 16 waves (128-VGPR cap)
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 2, 16, 392, 0, 1>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 2, 16, 392, 0, 1>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
#endif
    if (p.H == 392) {     // the model's size on the time path: compile-time geometry (7.2 -> 7.0 ms; the 32-row variant spills with it)
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 2, 16, 392>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 2, 16, 392>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
  if (upw <= 2) URSE_LB(2) else if (upw <= 3) URSE_LB(3) else if (upw <= 4) URSE_LB(4) else URSE_LB(6)
#undef URSE_LB
  URSE_CHECK_LAUNCH("urse_lstm_bwd");
  return URSE_OK;
  }
}

}  // namespace urse

using namespace urse;

static int check_map(const SeqMap& m, int H, int es, const char* who) {
  URSE_CHECK_ARG(m.n_seq > 0 && m.seq_len > 0 && m.inner > 0 && m.outer > 0 && m.stride > 0, "%s: bad sequence map",
                 who);
  URSE_CHECK_ARG(H > 0 && (4 * H * es) % 64 == 0, "%s: 4H*elemsize must be a multiple of 64 bytes (H=%d)", who, H);
  URSE_CHECK_ARG((H + 15) / 16 <= 48, "%s: H=%d too large (<= 768)", who, H);
  return URSE_OK;
}

extern "C" int urse_lstm_pack(const float* wih, const float* whh, const float* bih, const float* bhh, void* wih_p,
                              void* wihT_p, float* bias, void* whh_frag, void* whhT_frag, int N, int Np, int H, int Hp,
                              int dtype, void* stream) {
  URSE_CHECK_ARG(wih && whh && bih && bhh && wih_p && wihT_p && bias && whh_frag && whhT_frag, "urse_lstm_pack: null pointer");
  const int es = dtype == URSE_F32 ? 4 : 2;
  URSE_CHECK_ARG(dtype == URSE_F32 || dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_pack: bad dtype %d", dtype);
  URSE_CHECK_ARG(N > 0 && H > 0 && Np >= N && (Hp * es) % 64 == 0 && Hp >= ((H + 15) / 16) * 16 && (4 * H * es) % 64 == 0,
                 "urse_lstm_pack: bad shape N%d Np%d H%d Hp%d", N, Np, H, Hp);
  dim3 grid(256, 5), blk(256);
  if (dtype == URSE_F16)      // forward layouts f16, backward layouts (wihT, whhT) bf16
    hipLaunchKernelGGL((lstm_pack_kernel<f16_t, bf16_t>), grid, blk, 0, (hipStream_t)stream, wih, whh, bih, bhh, (f16_t*)wih_p,
                       (bf16_t*)wihT_p, bias, (f16_t*)whh_frag, (bf16_t*)whhT_frag, N, Np, H, Hp);
  else if (dtype == URSE_BF16)
    hipLaunchKernelGGL(lstm_pack_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, wih, whh, bih, bhh, (bf16_t*)wih_p,
                       (bf16_t*)wihT_p, bias, (bf16_t*)whh_frag, (bf16_t*)whhT_frag, N, Np, H, Hp);
  else
    hipLaunchKernelGGL(lstm_pack_kernel<float>, grid, blk, 0, (hipStream_t)stream, wih, whh, bih, bhh, (float*)wih_p,
                       (float*)wihT_p, bias, (float*)whh_frag, (float*)whhT_frag, N, Np, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_multi(const void* table, int n_lstm, int N, int Np, int H, int Hp, int dtype, void* stream) {
  const int es = dtype == URSE_F32 ? 4 : 2;
  URSE_CHECK_ARG(dtype == URSE_F32 || dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_pack_multi: bad dtype %d", dtype);
  URSE_CHECK_ARG(table && n_lstm > 0 && n_lstm < 65536 && N > 0 && H > 0 && Np >= N && (Hp * es) % 64 == 0 && Hp >= ((H + 15) / 16) * 16 &&
                     (4 * H * es) % 64 == 0, "urse_lstm_pack_multi: bad argument N%d Np%d H%d Hp%d", N, Np, H, Hp);
  dim3 grid(256, 5, (unsigned)n_lstm), blk(256);
  if (dtype == URSE_F16)
    hipLaunchKernelGGL((lstm_pack_multi_kernel<f16_t, bf16_t>), grid, blk, 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H, Hp);
  else if (dtype == URSE_BF16)
    hipLaunchKernelGGL(lstm_pack_multi_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H, Hp);
  else
    hipLaunchKernelGGL(lstm_pack_multi_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_multi");
  return URSE_OK;
}

extern "C" int urse_lstm_bidir_fwd(void* gx, int64_t ldg, const void* whh, void* hout, int64_t ldh, float* c, int H,
                                   int Hp, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                                   int save, int dtype, int rows16, void* hout_bf16, void* stream) {
  URSE_CHECK_ARG(gx && whh && hout && (c || !save), "urse_lstm_bidir_fwd: null pointer");
  URSE_CHECK_ARG(dtype == URSE_F32 || dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_bidir_fwd: bad dtype %d", dtype);
  URSE_CHECK_ARG(!hout_bf16 || dtype == URSE_F16, "urse_lstm_bidir_fwd: the bf16 copy of h goes with f16 operands only");
  LstmFwdArgs p;
  p.gx = gx; p.ldg = ldg; p.whh = whh; p.hout = hout; p.hout2 = hout_bf16; p.ldh = ldh; p.c = c; p.H = H; p.Hp = Hp; p.save = save;
  p.xcd = (xcd_dir_env() >> 2) & 1;
  p.m.inner = inner; p.m.outer = outer; p.m.stride = stride; p.m.n_seq = n_seq; p.m.seq_len = seq_len;
  const int es = dtype == URSE_F32 ? 4 : 2;
  int rc = check_map(p.m, H, es, "urse_lstm_bidir_fwd");
  if (rc) return rc;
  URSE_CHECK_ARG((Hp * es) % 64 == 0 && Hp >= ((H + 15) / 16) * 16, "urse_lstm_bidir_fwd: bad Hp %d for H %d", Hp, H);
  URSE_CHECK_ARG(ldg >= 8L * H && ldh >= 2L * H && ldg % 4 == 0, "urse_lstm_bidir_fwd: leading dimension too small");
  hipStream_t st = (hipStream_t)stream;
  // rows16: low 4 bits = row tiles per workgroup (0 auto), bit 4 = use 8 waves instead of 16 (tuning knob)
  int rt = rows16 & 15;
  bool nw8 = (rows16 >> 4) & 1;
  if (rt == 0) { rt = 1; nw8 = false; }
  note_launch(URSE_KV_LSTM_FWD_STREAM);
  if (dtype == URSE_F16) return launch_fwd<f16_t, 1, 16>(p, st);
  if (dtype == URSE_BF16) {
    if (nw8) {
      if (rt >= 4) return launch_fwd<bf16_t, 4, 8>(p, st);
      if (rt == 2) return launch_fwd<bf16_t, 2, 8>(p, st);
      return launch_fwd<bf16_t, 1, 8>(p, st);
    }
    if (rt >= 2 && H <= 256) return launch_fwd<bf16_t, 2, 16>(p, st);
    return launch_fwd<bf16_t, 1, 16>(p, st);
  }
  return launch_fwd<float, 1, 16>(p, st);
}

extern "C" int urse_lstm_bidir_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c,
                                   const void* whhT, int H, int n_seq, int seq_len, int64_t inner, int64_t outer,
                                   int64_t stride, int dtype, int rows16, void* stream) {
  URSE_CHECK_ARG(dh && gates && c && whhT, "urse_lstm_bidir_bwd: null pointer");
  LstmBwdArgs p;
  p.dh = dh; p.ldd = ldd; p.gates = gates; p.ldg = ldg; p.c = c; p.whhT = whhT; p.H = H;
  p.m.inner = inner; p.m.outer = outer; p.m.stride = stride; p.m.n_seq = n_seq; p.m.seq_len = seq_len;
  const int es = dtype == URSE_BF16 ? 2 : 4;
  int rc = check_map(p.m, H, es, "urse_lstm_bidir_bwd");
  if (rc) return rc;
  URSE_CHECK_ARG(ldg >= 8L * H && ldd >= 2L * H && ldg % 4 == 0, "urse_lstm_bidir_bwd: leading dimension too small");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldd < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_bidir_bwd: row indices must fit 32 bits");
  hipStream_t st = (hipStream_t)stream;
  int rt = rows16 & 15;
  bool nw8 = (rows16 >> 4) & 1;
  const bool fits2 = (size_t)32 * lds_frag_pitch(8 * H) <= 160 * 1024;
  if (rt == 0) {
    // many short sequences (band path): 32 sequences per workgroup of 8 waves halve the weight stream per sequence
    // (measured 3.9 vs 4.4 ms at C2); few long ones (time path) keep 16 per workgroup to fill the chip
    const bool many = (long)n_seq >= 32L * 128;
    rt = (many && fits2) ? 2 : 1;
    nw8 = many && fits2;
  }
  if (rt == 1 && (bwd_variant() == 2 || bwd_variant() == 6)) nw8 = true;
#ifdef URSE_EXPERIMENTS
  const bool fits3 = (size_t)48 * lds_frag_pitch(8 * H) + 48 * sizeof(int) <= 160 * 1024;
  if (dtype == URSE_BF16 && H == 392 && rt == 2 && fits3 && bwd_variant() == 8 && (ldg * 2) % 16 == 0 && ((uintptr_t)gates % 16) == 0) {
    note_launch(URSE_KV_LSTM_BWD_STREAM32);
    return launch_bwd<bf16_t, 3, 4>(p, st);
  }
#endif
  note_launch(dtype == URSE_BF16 && rt >= 2 && fits2 ? URSE_KV_LSTM_BWD_STREAM32 : URSE_KV_LSTM_BWD_STREAM16);
  if (dtype == URSE_BF16) {
    if (nw8) return (rt >= 2 && fits2) ? launch_bwd<bf16_t, 2, 8>(p, st) : launch_bwd<bf16_t, 1, 8>(p, st);
    if (rt >= 2 && fits2) return launch_bwd<bf16_t, 2, 16>(p, st);
    return launch_bwd<bf16_t, 1, 16>(p, st);
  }
  return launch_bwd<float, 1, 16>(p, st);
}

#ifdef BWSTAMP
extern "C" int urse_diag_bwd_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(urse::g_bwstamps), sizeof(unsigned long long) * 512 * 16);
}
#endif
