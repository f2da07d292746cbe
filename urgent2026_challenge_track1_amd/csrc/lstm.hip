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
  const void* whh;           // [2][4H][Hp] T
  void* hout; long ldh;      // [M, ldh] T: h (dir 0 cols [0,H), dir 1 cols [H,2H))
  float* c;                  // [M, 2H] f32 cell state (saved when `save`)
  int H, Hp, save;
  SeqMap m;
};

struct LstmBwdArgs {
  const void* dh; long ldd;  // [M, ldd] T: gradient w.r.t. hout
  void* gates; long ldg;     // in: saved gate activations; out: gradient w.r.t. gate pre-activations
  const float* c;            // [M, 2H]
  const void* whhT;          // [2][H][4H] T
  int H;
  SeqMap m;
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

constexpr int NW = 16;          // waves per workgroup: memory-level parallelism for the streamed weights
constexpr int NTHR = NW * 64;

template <typename T, int RT, int MAXUT>
__global__ void __launch_bounds__(NTHR) lstm_fwd_kernel(LstmFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T), R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  const int dir = blockIdx.y, s0 = blockIdx.x * R;
  const int H = p.H, Hp = p.Hp, nut = (H + 15) >> 4;
  const int pitch = Hp * ES + 16;
  for (int i = tid; i < 2 * R * pitch / 4; i += NTHR) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  float cst[MAXUT][RT][4];
#pragma unroll
  for (int a = 0; a < MAXUT; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) cst[a][b][c] = 0.f;
  long rowbase[RT][4];
  bool rvalid[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      rvalid[rt][r] = seq < p.m.n_seq;
      if (seq >= p.m.n_seq) seq = p.m.n_seq - 1;
      rowbase[rt][r] = (seq / p.m.inner) * p.m.outer + (seq % p.m.inner);
    }
  const char* whh = reinterpret_cast<const char*>(p.whh) + (long)dir * 4 * H * Hp * ES;
  T* gx = reinterpret_cast<T*>(p.gx);
  T* hout = reinterpret_cast<T*>(p.hout);
  const int nslab = Hp * ES / 64;
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
        f32x4_t acc[4][RT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[g][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* wr = whh + ((long)uc * Hp) * ES + 16 * lr;
        const long wg = (long)H * Hp * ES;  // gate stride
        const char* ar = hc + lc * pitch + 16 * lr;
#pragma unroll 4
        for (int ks = 0; ks < nslab; ++ks) {
          uint4 b[4], a[RT];
#pragma unroll
          for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const uint4*>(wr + g * wg + ks * 64);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + ks * 64);
#pragma unroll
          for (int g = 0; g < 4; ++g) mma_slab<T, RT>(a, b[g], acc[g]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long row = rowbase[rt][r] + toff;
            T* gp = gx + row * p.ldg + gcol0 + uc;
            const float gi = acc[0][rt][r] + to_f32<T>(gp[0]);
            const float gf = acc[1][rt][r] + to_f32<T>(gp[H]);
            const float gg = acc[2][rt][r] + to_f32<T>(gp[2 * H]);
            const float go = acc[3][rt][r] + to_f32<T>(gp[3 * H]);
            const float iv = sigmoidf_(gi), fv = sigmoidf_(gf), gv = tanhf_(gg), ov = sigmoidf_(go);
            const float cv = fv * cst[ui][rt][r] + iv * gv;
            cst[ui][rt][r] = cv;
            const float hv = uvalid ? ov * tanhf_(cv) : 0.f;
            const T hT = from_f32<T>(hv);
            *reinterpret_cast<T*>(hn + (rt * 16 + lr * 4 + r) * pitch + u * ES) = hT;
            if (rvalid[rt][r] && uvalid) {
              hout[row * p.ldh + (long)dir * H + u] = hT;
              if (p.save) {
                gp[0] = from_f32<T>(iv);
                gp[H] = from_f32<T>(fv);
                gp[2 * H] = from_f32<T>(gv);
                gp[3 * H] = from_f32<T>(ov);
                p.c[row * 2 * H + (long)dir * H + u] = cv;
              }
            }
          }
      }
    }
    __syncthreads();
  }
}

template <typename T, int RT, int MAXUT>
__global__ void __launch_bounds__(NTHR) lstm_bwd_kernel(LstmBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ES = sizeof(T), R = 16 * RT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  const int dir = blockIdx.y, s0 = blockIdx.x * R;
  const int H = p.H, nut = (H + 15) >> 4, G4 = 4 * H;
  const int pitch = G4 * ES + 16;
  float dcs[MAXUT][RT][4], dhr[MAXUT][RT][4];
#pragma unroll
  for (int a = 0; a < MAXUT; ++a)
#pragma unroll
    for (int b = 0; b < RT; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) { dcs[a][b][c] = 0.f; dhr[a][b][c] = 0.f; }
  long rowbase[RT][4];
  bool rvalid[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      rvalid[rt][r] = seq < p.m.n_seq;
      if (seq >= p.m.n_seq) seq = p.m.n_seq - 1;
      rowbase[rt][r] = (seq / p.m.inner) * p.m.outer + (seq % p.m.inner);
    }
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + (long)dir * H * G4 * ES;
  const T* dh = reinterpret_cast<const T*>(p.dh);
  T* gates = reinterpret_cast<T*>(p.gates);
  const int nslab = G4 * ES / 64;
  const long gcol0 = (long)dir * G4;
  const long prev_off = dir ? p.m.stride : -p.m.stride;

  for (int step = 0; step < p.m.seq_len; ++step) {
    const int t = dir ? step : (p.m.seq_len - 1 - step);
    const bool first = dir ? (t == p.m.seq_len - 1) : (t == 0);  // first step of the forward recurrence
    const long toff = (long)t * p.m.stride;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = w + NW * ui;
      if (ut < nut) {
        const int u = ut * 16 + lc;
        if (u < H) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const long row = rowbase[rt][r] + toff;
              T* gp = gates + row * p.ldg + gcol0 + u;
              const float iv = to_f32<T>(gp[0]), fv = to_f32<T>(gp[H]), gv = to_f32<T>(gp[2 * H]),
                          ov = to_f32<T>(gp[3 * H]);
              const long ci = row * 2 * H + (long)dir * H + u;
              const float ct = p.c[ci];
              const float cp = first ? 0.f : p.c[ci + prev_off * 2 * H];
              const float dhv = to_f32<T>(dh[row * p.ldd + (long)dir * H + u]) + dhr[ui][rt][r];
              const float tc = tanhf_(ct);
              const float dct = dcs[ui][rt][r] + dhv * ov * (1.f - tc * tc);
              const float dgo = dhv * tc * ov * (1.f - ov);
              const float dgi = dct * gv * iv * (1.f - iv);
              const float dgf = dct * cp * fv * (1.f - fv);
              const float dgg = dct * iv * (1.f - gv * gv);
              dcs[ui][rt][r] = dct * fv;
              const T ti = from_f32<T>(dgi), tf = from_f32<T>(dgf), tg = from_f32<T>(dgg), to = from_f32<T>(dgo);
              char* lrow = smem + (rt * 16 + lr * 4 + r) * pitch;
              *reinterpret_cast<T*>(lrow + (u)*ES) = ti;
              *reinterpret_cast<T*>(lrow + (H + u) * ES) = tf;
              *reinterpret_cast<T*>(lrow + (2 * H + u) * ES) = tg;
              *reinterpret_cast<T*>(lrow + (3 * H + u) * ES) = to;
              if (rvalid[rt][r]) {
                gp[0] = ti;
                gp[H] = tf;
                gp[2 * H] = tg;
                gp[3 * H] = to;
              }
            }
        }
      }
    }
    __syncthreads();
    if (step + 1 < p.m.seq_len) {
#pragma unroll
      for (int ui = 0; ui < MAXUT; ++ui) {
        const int ut = w + NW * ui;
        if (ut < nut) {
          const int u = ut * 16 + lc;
          const int uc = u < H ? u : H - 1;
          f32x4_t acc[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          const char* wr = whhT + ((long)uc * G4) * ES + 16 * lr;
          const char* ar = smem + lc * pitch + 16 * lr;
#pragma unroll 4
          for (int ks = 0; ks < nslab; ++ks) {
            uint4 a[RT];
            const uint4 b = *reinterpret_cast<const uint4*>(wr + ks * 64);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = *reinterpret_cast<const uint4*>(ar + rt * 16 * pitch + ks * 64);
            mma_slab<T, RT>(a, b, acc);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhr[ui][rt][r] = acc[rt][r];
        }
      }
    }
    __syncthreads();
  }
}

template <typename K>
static void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}

template <typename T, int RT, int MAXUT>
static int launch_fwd(const LstmFwdArgs& p, hipStream_t st) {
  static bool once = (allow_big_lds(lstm_fwd_kernel<T, RT, MAXUT>), true);
  (void)once;
  const int R = 16 * RT;
  const size_t lds = (size_t)2 * R * (p.Hp * sizeof(T) + 16);
  URSE_CHECK_ARG(lds <= 160 * 1024, "urse_lstm_fwd: Hp %d with %d rows exceeds LDS", p.Hp, R);
  dim3 grid(ceil_div(p.m.n_seq, R), 2);
  hipLaunchKernelGGL((lstm_fwd_kernel<T, RT, MAXUT>), grid, dim3(NTHR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_fwd");
  return URSE_OK;
}

template <typename T, int RT, int MAXUT>
static int launch_bwd(const LstmBwdArgs& p, hipStream_t st) {
  static bool once = (allow_big_lds(lstm_bwd_kernel<T, RT, MAXUT>), true);
  (void)once;
  const int R = 16 * RT;
  const size_t lds = (size_t)R * (4 * p.H * sizeof(T) + 16);
  URSE_CHECK_ARG(lds <= 160 * 1024, "urse_lstm_bwd: H %d with %d rows exceeds LDS", p.H, R);
  dim3 grid(ceil_div(p.m.n_seq, R), 2);
  hipLaunchKernelGGL((lstm_bwd_kernel<T, RT, MAXUT>), grid, dim3(NTHR), lds, st, p);
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

extern "C" int urse_lstm_bidir_fwd(void* gx, int64_t ldg, const void* whh, void* hout, int64_t ldh, float* c, int H,
                                   int Hp, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                                   int save, int dtype, int rows16, void* stream) {
  URSE_CHECK_ARG(gx && whh && hout && (c || !save), "urse_lstm_bidir_fwd: null pointer");
  LstmFwdArgs p;
  p.gx = gx; p.ldg = ldg; p.whh = whh; p.hout = hout; p.ldh = ldh; p.c = c; p.H = H; p.Hp = Hp; p.save = save;
  p.m.inner = inner; p.m.outer = outer; p.m.stride = stride; p.m.n_seq = n_seq; p.m.seq_len = seq_len;
  const int es = dtype == URSE_BF16 ? 2 : 4;
  int rc = check_map(p.m, H, es, "urse_lstm_bidir_fwd");
  if (rc) return rc;
  URSE_CHECK_ARG((Hp * es) % 64 == 0 && Hp >= ((H + 15) / 16) * 16, "urse_lstm_bidir_fwd: bad Hp %d for H %d", Hp, H);
  URSE_CHECK_ARG(ldg >= 8L * H && ldh >= 2L * H, "urse_lstm_bidir_fwd: leading dimension too small");
  const int ut_per_wave = ((H + 15) / 16 + NW - 1) / NW;
  hipStream_t st = (hipStream_t)stream;
  int rt = rows16;
  if (rt <= 0) rt = n_seq >= 1024 ? 2 : 1;
  if (dtype == URSE_BF16) {
    if (ut_per_wave <= 2) return rt >= 2 ? launch_fwd<bf16_t, 2, 2>(p, st) : launch_fwd<bf16_t, 1, 2>(p, st);
    return rt >= 2 ? launch_fwd<bf16_t, 2, 3>(p, st) : launch_fwd<bf16_t, 1, 3>(p, st);
  }
  if (ut_per_wave <= 2) return launch_fwd<float, 1, 2>(p, st);
  return launch_fwd<float, 1, 3>(p, st);
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
  URSE_CHECK_ARG(ldg >= 8L * H && ldd >= 2L * H, "urse_lstm_bidir_bwd: leading dimension too small");
  const int ut_per_wave = ((H + 15) / 16 + NW - 1) / NW;
  hipStream_t st = (hipStream_t)stream;
  int rt = rows16;
  if (rt <= 0) rt = n_seq >= 1024 ? 2 : 1;
  if (dtype == URSE_BF16) {
    const bool fits2 = (size_t)32 * (8 * H + 16) <= 160 * 1024;
    if (ut_per_wave <= 2) return (rt >= 2 && fits2) ? launch_bwd<bf16_t, 2, 2>(p, st) : launch_bwd<bf16_t, 1, 2>(p, st);
    return launch_bwd<bf16_t, 1, 3>(p, st);
  }
  if (ut_per_wave <= 2) return launch_bwd<float, 1, 2>(p, st);
  return launch_bwd<float, 1, 3>(p, st);
}
