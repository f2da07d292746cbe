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
template <> struct Vec4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ void unpack(raw v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ raw pack(const float (&i)[4]) { return make_float4(i[0], i[1], i[2], i[3]); }
};

template <typename T, int RT>
__device__ __forceinline__ void mma_slab(const uint4 (&a)[RT], const uint4& b, f32x4_t (&acc)[RT]) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[rt]),
                                                        __builtin_bit_cast(bf16x8_t, b), acc[rt], 0, 0, 0);
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
              if (p.save) {
                const float act[4] = {iv, fv, gv, ov};
                *reinterpret_cast<V4*>(gx + row * p.ldg + gcol0 + u * 4) = Vec4<T>::pack(act);
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
template <typename T, int RT, int MAXUT, int NW, int HC = 0, int HPC = 0>
__global__ void __launch_bounds__(NW * 64) lstm_bwd_kernel(LstmBwdArgs p) {
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
  const int pitch = HPC ? lds_frag_pitch(4 * HPC * ES) : lds_frag_pitch(G4 * ES);
  const int nbuf = p.dbuf ? 2 : 1;         // double-buffered dgates tile: one barrier per step
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
  const int nslab = G4 * ES / 64;
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * nut * nslab) * 1024 + lane * 16;
  const T* dh = reinterpret_cast<const T*>(p.dh);
  T* gates = reinterpret_cast<T*>(p.gates);
  // 32-bit row indices / leading dimensions (checked on the host): an address costs one v_mad_i64_i32
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.m.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;

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
  for (int step = 0; step < p.m.seq_len; ++step) {
    const int t = dir ? step : (p.m.seq_len - 1 - step);
    const bool first = dir ? (t == p.m.seq_len - 1) : (t == 0);   // first step of the forward recurrence: c_{-1} = 0
    char* tile = smem + (step % nbuf) * R * pitch;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        const int u = ut * 16 + lc;
        if (u < H) {
          V4 gpre[RT][4];
          float cpre[RT][4];
          T dhpre[RT][4];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = (int)rowb(rt, r) + t * stride_i;
#ifdef BABL_NO_P1LOAD
              gpre[rt][r] = V4{}; cpre[rt][r] = (float)row; dhpre[rt][r] = T(row & 1);
#else
              gpre[rt][r] = *reinterpret_cast<const V4*>(gates + ((long)row * ldg_i + (gcol_i + u * 4)));
              cpre[rt][r] = first ? 0.f : p.c[(long)(row + prev_i) * ldc_i + (hcol_i + u)];
              dhpre[rt][r] = dh[(long)row * ldd_i + (hcol_i + u)];
#endif
            }
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
              *reinterpret_cast<V4*>(tile + (rt * 16 + lr * 4 + r) * pitch + (u * 4) * ES) = pk;
#ifndef BABL_NO_STORE
              if (rowbase[rt][r] >= 0)
                *reinterpret_cast<V4*>(gates + ((long)(rowbase[rt][r] + t * stride_i) * ldg_i + (gcol_i + u * 4))) = pk;
#endif
            }
        }
      }
    }
    if (step + 1 == p.m.seq_len) break;
    __syncthreads();
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
        constexpr int KB = (sizeof(T) == 2) ? (RT >= 2 ? URSE_BWD_KB2 : URSE_BWD_KB) : 8;
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
template <typename T>
__global__ void __launch_bounds__(256) lstm_pack_kernel(const float* __restrict__ wih, const float* __restrict__ whh,
                                                        const float* __restrict__ bih, const float* __restrict__ bhh,
                                                        T* __restrict__ wih_p, T* __restrict__ wihT_p,
                                                        float* __restrict__ bias, T* __restrict__ whh_f,
                                                        T* __restrict__ whhT_f, int N, int Np, int H, int Hp) {
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
      wihT_p[idx] = from_f32<T>(wih[((long)d * G4 + g * H + u) * N + n]);
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
      whhT_f[idx] = from_f32<T>(n < H ? whh[((long)d * G4 + gp * H + up) * H + n] : 0.f);
    }
  }
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

template <typename T, int RT, int NW>
static int launch_bwd(const LstmBwdArgs& p, hipStream_t st) {
  constexpr int R = 16 * RT;
  size_t lds = (size_t)R * lds_frag_pitch(4 * p.H * (int)sizeof(T));
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
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 2, 4, 8, 0, 392>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 2, 4, 8, 0, 392>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
  if constexpr (sizeof(T) == 2 && NW == 16 && RT == 1) {
    if (p.H == 392) {     // the model's size on the time path: compile-time geometry (7.2 -> 7.0 ms; the 32-row variant spills with it)
      static bool once = (allow_big_lds(lstm_bwd_kernel<T, 1, 2, 16, 392>), true);
      (void)once;
      hipLaunchKernelGGL((lstm_bwd_kernel<T, 1, 2, 16, 392>), grid, dim3(NW * 64), lds, st, pa);
      URSE_CHECK_LAUNCH("urse_lstm_bwd");
      return URSE_OK;
    }
  }
  if (upw <= 2) URSE_LB(2) else if (upw <= 4) URSE_LB(4) else URSE_LB(6)
#undef URSE_LB
  URSE_CHECK_LAUNCH("urse_lstm_bwd");
  return URSE_OK;
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
  const int es = dtype == URSE_BF16 ? 2 : 4;
  URSE_CHECK_ARG(N > 0 && H > 0 && Np >= N && (Hp * es) % 64 == 0 && Hp >= ((H + 15) / 16) * 16 && (4 * H * es) % 64 == 0,
                 "urse_lstm_pack: bad shape N%d Np%d H%d Hp%d", N, Np, H, Hp);
  dim3 grid(256, 5), blk(256);
  if (dtype == URSE_BF16)
    hipLaunchKernelGGL(lstm_pack_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, wih, whh, bih, bhh, (bf16_t*)wih_p,
                       (bf16_t*)wihT_p, bias, (bf16_t*)whh_frag, (bf16_t*)whhT_frag, N, Np, H, Hp);
  else
    hipLaunchKernelGGL(lstm_pack_kernel<float>, grid, blk, 0, (hipStream_t)stream, wih, whh, bih, bhh, (float*)wih_p,
                       (float*)wihT_p, bias, (float*)whh_frag, (float*)whhT_frag, N, Np, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack");
  return URSE_OK;
}

extern "C" int urse_lstm_bidir_fwd(void* gx, int64_t ldg, const void* whh, void* hout, int64_t ldh, float* c, int H,
                                   int Hp, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                                   int save, int dtype, int rows16, void* stream) {
  URSE_CHECK_ARG(gx && whh && hout && (c || !save), "urse_lstm_bidir_fwd: null pointer");
  LstmFwdArgs p;
  p.gx = gx; p.ldg = ldg; p.whh = whh; p.hout = hout; p.ldh = ldh; p.c = c; p.H = H; p.Hp = Hp; p.save = save;
  p.xcd = (xcd_dir_env() >> 2) & 1;
  p.m.inner = inner; p.m.outer = outer; p.m.stride = stride; p.m.n_seq = n_seq; p.m.seq_len = seq_len;
  const int es = dtype == URSE_BF16 ? 2 : 4;
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
  note_launch(dtype == URSE_BF16 && rt >= 2 && fits2 ? URSE_KV_LSTM_BWD_STREAM32 : URSE_KV_LSTM_BWD_STREAM16);
  if (dtype == URSE_BF16) {
    if (nw8) return (rt >= 2 && fits2) ? launch_bwd<bf16_t, 2, 8>(p, st) : launch_bwd<bf16_t, 1, 8>(p, st);
    if (rt >= 2 && fits2) return launch_bwd<bf16_t, 2, 16>(p, st);
    return launch_bwd<bf16_t, 1, 16>(p, st);
  }
  return launch_bwd<float, 1, 16>(p, st);
}
