// Framed STFT / iSTFT for gfx950: frames are staged in LDS, transformed by the mixed-radix
// Stockham engine of fft_lds.h (two real frames per complex FFT) and written back coalesced.
// HBM-bound by design: one read of the waveform span, one write of the [T,F] spectra (or the
// reverse); the 50 %/75 % frame overlap is served from LDS / L2, never re-read from HBM.
//
// Replaces torch.stft / torch.istft behind espnet2 Stft.forward / Stft.inverse
// (reference call sites: baseline_code/models/bsrnn.py:37,40; flow_model.py:136,145).
#include <map>
#include <mutex>
#include <vector>
#include <math.h>
#include <stdlib.h>

#include "fft_lds.h"

namespace urse {

struct StftTables {
  FftPlan plan;
  float2* tw;     // device, n
  float* win[2];  // device, n  (rect, hann)
};

static std::mutex g_mu;
static std::map<std::pair<int, int>, StftTables> g_tables;  // (device, n_fft)

static int get_tables(int n, StftTables* out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find({dev, n});
  if (it != g_tables.end()) { *out = it->second; return URSE_OK; }
  StftTables t;
  if (!make_fft_plan(n, &t.plan)) {
    set_error("stft: n_fft=%d has a prime factor > 61 or too many factors", n);
    return URSE_ERR_UNSUPPORTED;
  }
  std::vector<float2> tw(n);
  std::vector<float> w0(n, 1.0f), w1(n);
  for (int j = 0; j < n; ++j) {
    const double a = -2.0 * M_PI * (double)j / (double)n;
    tw[j] = make_float2((float)cos(a), (float)sin(a));
    w1[j] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)j / (double)n));  // periodic Hann
  }
  if (hipMalloc(&t.tw, n * sizeof(float2)) != hipSuccess || hipMalloc(&t.win[0], n * sizeof(float)) != hipSuccess ||
      hipMalloc(&t.win[1], n * sizeof(float)) != hipSuccess) {
    set_error("stft: hipMalloc of plan tables failed");
    return URSE_ERR_RUNTIME;
  }
  (void)hipMemcpy(t.tw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice);
  (void)hipMemcpy(t.win[0], w0.data(), n * sizeof(float), hipMemcpyHostToDevice);
  (void)hipMemcpy(t.win[1], w1.data(), n * sizeof(float), hipMemcpyHostToDevice);
  g_tables[{dev, n}] = t;
  *out = t;
  return URSE_OK;
}

// number of complex FFTs (frame pairs) a workgroup carries so that two workgroups fit a CU's LDS
static int pick_nf(int n, int min_nf) {
  // measured (scripts/time_stft.py, B32 x 4 s @ 48 kHz): the kernels are VALU-issue bound, one frame pair per 256-thread
  // workgroup is fastest (NF 1/2/4/8 -> 53/64/74/129 us for the 960-point STFT)
  int nf = 1;
  if (const char* e = getenv("URSE_STFT_NF")) { nf = atoi(e); return nf < min_nf ? min_nf : nf; }   // tuning knob
  while (nf > min_nf && (size_t)(2 * nf * n) * sizeof(float2) + n * 12 > 76 * 1024) --nf;
  return nf < min_nf ? min_nf : nf;
}
static size_t lds_bytes(int n, int nf) { return (size_t)n * 8 + (size_t)n * 4 + (size_t)2 * nf * n * 8; }
static int stft_threads() {
  if (const char* e = getenv("URSE_STFT_THREADS")) return atoi(e);
  return 256;
}

// dynamic LDS above 64 KiB must be opted into once per kernel
template <typename K>
static void allow_big_lds(K kernel) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
}
static std::once_flag g_lds_once;

// sum_t w^2[m - t*hop] over the frames 0 <= t < T that cover padded position m
__device__ __forceinline__ float ola_envelope(const float* win, int m, int n, int hop, int T) {
  int thi = m / hop;
  if (thi > T - 1) thi = T - 1;
  int tlo = (m - n + hop) / hop;  // ceil((m-n+1)/hop) for m-n+1 > 0
  if (m - n + 1 <= 0) tlo = 0;
  float e = 0.f;
  for (int t = tlo; t <= thi; ++t) {
    const float w = win[m - t * hop];
    e += w * w;
  }
  return e;
}

// MODE 0: STFT (reflect padding).  MODE 1: adjoint of iSTFT (zero padding, input divided by the
// OLA envelope, bins scaled by c_k/n with c_k = 1 for DC/Nyquist and 2 otherwise).
template <int MODE>
__global__ void __launch_bounds__(256) stft_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                   float2* __restrict__ out, int L, int T, FftPlan plan, int hop,
                                                   const float* __restrict__ win_g, const float2* __restrict__ tw_g,
                                                   int NF) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = plan.n;
  float2* tw = reinterpret_cast<float2*>(smem);
  float* win = reinterpret_cast<float*>(smem + (size_t)n * 8);
  float2* bufA = reinterpret_cast<float2*>(smem + (size_t)n * 12);
  float2* bufB = bufA + (size_t)NF * n;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 2 * NF;
  const int half = n / 2;
  const int F = half + 1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    tw[i] = tw_g[i];
    win[i] = win_g[i];
  }
  if (MODE == 1) __syncthreads();  // envelope needs the window
  const float* xb = x + (size_t)b * L;
  for (int idx = threadIdx.x; idx < NF * n; idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_n), i = idx - f * n;
    float v[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t = t0 + 2 * f + s;
      float val = 0.f;
      if (t < T) {
        const int m = t * hop + i;
        int p = m - half;
        if (MODE == 0) {
          if (p < 0) p = -p;
          if (p >= L) p = 2 * (L - 1) - p;
          val = xb[p];
        } else {
          if (p >= 0 && p < L) val = xb[p] / ola_envelope(win, m, n, hop, T);
        }
      }
      v[s] = val;
    }
    const float w = (MODE == 0) ? win_g[i] : win[i];
    bufA[idx] = make_float2(v[0] * w, v[1] * w);
  }
  __syncthreads();
  const float2* Z = fft_lds_forward(bufA, bufB, NF, plan, tw);
  int olen = T;
  if (MODE == 0 && lens != nullptr) olen = (lens[b] + 2 * half - n) / hop + 1;
  const bool even = (n & 1) == 0;
  const float inv_n = 1.0f / (float)n;
  for (int idx = threadIdx.x; idx < NF * F; idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_f), k = idx - f * F;
    const float2 zk = Z[f * n + k];
    const float2 zc = Z[f * n + (k == 0 ? 0 : n - k)];
    // Xa = (zk + conj(zc))/2 ; Xb = -i (zk - conj(zc))/2
    float2 xa = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
    float2 xbv = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
    if (MODE == 1) {
      const bool edge = (k == 0) || (even && k == half);
      const float s = edge ? inv_n : 2.0f * inv_n;
      xa.x *= s; xbv.x *= s;
      xa.y = edge ? 0.f : xa.y * s;
      xbv.y = edge ? 0.f : xbv.y * s;
    }
    const int ta = t0 + 2 * f;
    if (ta < T) out[((size_t)b * T + ta) * F + k] = (ta < olen) ? xa : make_float2(0.f, 0.f);
    if (ta + 1 < T) out[((size_t)b * T + ta + 1) * F + k] = (ta + 1 < olen) ? xbv : make_float2(0.f, 0.f);
  }
}

#ifndef URSE_STFT960_NFF
#define URSE_STFT960_NFF 8
#endif
#define URSE_STFT960_NFF_THREADS (URSE_STFT960_NFF * 32)
// ---------------------------------------------------------------------------------------------------------------
// 960-point forward STFT (48 kHz: n_fft 960 / hop 480, the C2 front end), register FFT.
// The generic kernel above walks five radix passes through LDS with run-time index arithmetic and is VALU-issue bound
// (52 us for 74 MB).  Here one half-wave (32 lanes) owns one complex FFT = two real frames, as a 32 x 30 two-pass FFT:
//   pass 1: lane n2 (30 lanes) loads x[30 n1 + n2], n1 = 0..31 (lanes read consecutive samples), and does a 32-point
//           DFT in registers (radix-2 DIF, constant twiddles), multiplies by W_960^(n2 k1) and writes B[k1][n2] to LDS;
//   pass 2: lane k1 (32 lanes) reads B[k1][0..29] and does a 30-point DFT in registers as a 2 x 3 x 5 prime-factor
//           transform (Good's index map: no twiddles), X[k1 + 32 k2], written back to LDS in natural order;
//   output: the two real frames' spectra are split out of the complex one (X[k], conj X[960-k]) and stored coalesced.
// Two LDS round trips instead of five, index arithmetic folded at compile time.  8 FFTs (16 frames) per workgroup.
__device__ __forceinline__ void dft32_dif(float2 (&v)[32]) {
  // forward DFT, natural order in, bit-reversed order out: X[k] = v[brev5(k)]
  constexpr float C32[16] = {1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f,
                             0.382683432f, 0.195090322f, 0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f,
                             -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f};
  constexpr float S32[16] = {0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f,
                             0.923879533f, 0.980785280f, 1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f,
                             0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    const int half = 16 >> s;
#pragma unroll
    for (int blk = 0; blk < (1 << s); ++blk)
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const int i0 = blk * 2 * half + j, i1 = i0 + half;
        const float2 a = v[i0], b = v[i1];
        v[i0] = make_float2(a.x + b.x, a.y + b.y);
        const float dx = a.x - b.x, dy = a.y - b.y;
        const int m = j << s;                               // twiddle W_32^m = cos - i sin
        if (m == 0) v[i1] = make_float2(dx, dy);
        else if (m == 8) v[i1] = make_float2(dy, -dx);
        else v[i1] = make_float2(dx * C32[m] + dy * S32[m], dy * C32[m] - dx * S32[m]);
      }
  }
}

__device__ __forceinline__ void dft3_inplace(float2& v0, float2& v1, float2& v2) {
  const float sn = 0.86602540378443864676f;
  const float2 sm = make_float2(v1.x + v2.x, v1.y + v2.y), d = make_float2(v1.x - v2.x, v1.y - v2.y);
  const float2 m = make_float2(v0.x - 0.5f * sm.x, v0.y - 0.5f * sm.y);
  const float2 r = make_float2(sn * d.y, -sn * d.x);
  v0 = make_float2(v0.x + sm.x, v0.y + sm.y);
  v1 = make_float2(m.x + r.x, m.y + r.y);
  v2 = make_float2(m.x - r.x, m.y - r.y);
}

__device__ __forceinline__ void dft5_inplace(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const float2 a14 = make_float2(v1.x + v4.x, v1.y + v4.y), b14 = make_float2(v1.x - v4.x, v1.y - v4.y);
  const float2 a23 = make_float2(v2.x + v3.x, v2.y + v3.y), b23 = make_float2(v2.x - v3.x, v2.y - v3.y);
  const float2 p1 = make_float2(v0.x + c1 * a14.x + c2 * a23.x, v0.y + c1 * a14.y + c2 * a23.y);
  const float2 p2 = make_float2(v0.x + c2 * a14.x + c1 * a23.x, v0.y + c2 * a14.y + c1 * a23.y);
  const float2 q1 = make_float2(s1 * b14.y + s2 * b23.y, -(s1 * b14.x + s2 * b23.x));
  const float2 q2 = make_float2(s2 * b14.y - s1 * b23.y, -(s2 * b14.x - s1 * b23.x));
  v0 = make_float2(v0.x + a14.x + a23.x, v0.y + a14.y + a23.y);
  v1 = make_float2(p1.x + q1.x, p1.y + q1.y);
  v4 = make_float2(p1.x - q1.x, p1.y - q1.y);
  v2 = make_float2(p2.x + q2.x, p2.y + q2.y);
  v3 = make_float2(p2.x - q2.x, p2.y - q2.y);
}

// 30-point forward DFT in place as a 2 x 3 x 5 prime-factor transform: element (a, b, c) lives at index
// (15a + 10b + 6c) mod 30 on input AND output (the CRT and Good maps coincide for 30 = 2*3*5)
__device__ __forceinline__ void dft30_pfa(float2 (&u)[30]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int base = 15 * a + 10 * b;
      dft5_inplace(u[base % 30], u[(base + 6) % 30], u[(base + 12) % 30], u[(base + 18) % 30], u[(base + 24) % 30]);
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int base = 15 * a + 6 * c;
      dft3_inplace(u[base % 30], u[(base + 10) % 30], u[(base + 20) % 30]);
    }
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int i0 = (10 * b + 6 * c) % 30, i1 = (i0 + 15) % 30;
      const float2 x0 = u[i0], x1 = u[i1];
      u[i0] = make_float2(x0.x + x1.x, x0.y + x1.y);
      u[i1] = make_float2(x0.x - x1.x, x0.y - x1.y);
    }
}

// MODE 0: STFT (reflect padding).  MODE 1: adjoint of the hop-480 iSTFT (zero padding, input divided by the OLA envelope, bins
// scaled by c_k / n), as stft_kernel<1>.
template <int MODE>
__global__ void __launch_bounds__(URSE_STFT960_NFF_THREADS) stft960_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                      float2* __restrict__ out, int L, int T, int hop,
                                                      int hann, const float2* __restrict__ tw_g) {
#ifndef URSE_STFT960_NFF
#define URSE_STFT960_NFF 8
#endif
  constexpr int N = 960, F = 481, NFF = URSE_STFT960_NFF;   // complex FFTs (frame pairs) per workgroup
  constexpr int NTH = NFF * 32;
  constexpr int ZS = 1000;                                  // per-FFT LDS stride: 8000 B = 16 banks of skew between half-waves
  __shared__ float2 zbuf[NFF][ZS];
  __shared__ float2 tw[N];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31;
  const int f = 2 * (tid >> 6) + (lane >> 5);
  const int b = blockIdx.y;
  const int ta = blockIdx.x * 2 * NFF + 2 * f;
  for (int i = tid; i < N; i += NTH) tw[i] = tw_g[i];
  const float* xb = x + (size_t)b * L;
  float2 v[32];
  // branch-free loads (frames past T are clamped to a valid frame and multiplied by 0): all samples of a lane are in
  // flight together.  With hop == 480 == 30 * 16 the second frame of the pair is the first one shifted by 16 rows of the
  // 32 x 30 sample matrix, so only its last 16 rows are loaded.
  const bool half_hop = hop * 2 == N;
  const int tac = ta < T ? ta : T - 1, tbc = ta + 1 < T ? ta + 1 : T - 1;
  const float ma = ta < T ? 1.f : 0.f, mb = ta + 1 < T ? 1.f : 0.f;
  const int lq = l < 30 ? l : 29;
  const int pa0 = tac * hop - N / 2 + lq, pb0 = tbc * hop - N / 2 + lq;
  float xa_[32], xb_[32];
  if (MODE == 1) {
    // zero padding; the second frame of the pair shares its first 16 rows with the first one (hop 480 only)
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
      const int qa = pa0 + 30 * n1;
      const bool ok = qa >= 0 && qa < L;
      const float val = xb[ok ? qa : 0];
      xa_[n1] = ok ? val : 0.f;
    }
    if (tbc == tac + 1) {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) xb_[n1] = xa_[n1 + 16];
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int qb = pb0 + 30 * n1;
        const bool ok = qb >= 0 && qb < L;
        const float val = xb[ok ? qb : 0];
        xb_[n1] = ok ? val : 0.f;
      }
    }
#pragma unroll
    for (int n1 = 16; n1 < 32; ++n1) {
      const int qb = pb0 + 30 * n1;
      const bool ok = qb >= 0 && qb < L;
      const float val = xb[ok ? qb : 0];
      xb_[n1] = ok ? val : 0.f;
    }
  } else {
#pragma unroll
  for (int n1 = 0; n1 < 32; ++n1) {
    int qa = pa0 + 30 * n1;
    qa = qa < 0 ? -qa : qa;
    qa = qa >= L ? 2 * (L - 1) - qa : qa;
    xa_[n1] = xb[qa];
  }
  if (half_hop && tbc == tac + 1) {
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) xb_[n1] = xa_[n1 + 16];
#pragma unroll
    for (int n1 = 16; n1 < 32; ++n1) {
      int qb = pb0 + 30 * n1;
      qb = qb >= L ? 2 * (L - 1) - qb : qb;            // (the second frame's last rows never fall before sample 0)
      xb_[n1] = xb[qb];
    }
  } else {
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
      int qb = pb0 + 30 * n1;
      qb = qb < 0 ? -qb : qb;
      qb = qb >= L ? 2 * (L - 1) - qb : qb;
      xb_[n1] = xb[qb];
    }
  }
  }
  __syncthreads();                                          // twiddle table complete
  if (l < 30) {
    // window: periodic Hann = 0.5 - 0.5 cos(2 pi i / N) = 0.5 - 0.5 Re(W^i) straight from the twiddle table in LDS
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
      const int i = 30 * n1 + l;
      const float w = hann ? 0.5f - 0.5f * tw[i].x : 1.0f;
      if (MODE == 1) {
        // OLA envelope at padded position m = t * 480 + i: frames m / 480 (sample r) and m / 480 - 1 (sample r + 480)
        const int r = i < 480 ? i : i - 480;
        const float w0 = hann ? 0.5f - 0.5f * tw[r].x : 1.0f, w1 = hann ? 0.5f - 0.5f * tw[r + 480].x : 1.0f;
        const int ja = tac + (i >= 480 ? 1 : 0), jb = tbc + (i >= 480 ? 1 : 0);
        const float ea = (ja < T ? w0 * w0 : 0.f) + (ja >= 1 ? w1 * w1 : 0.f);
        const float eb = (jb < T ? w0 * w0 : 0.f) + (jb >= 1 ? w1 * w1 : 0.f);
        xa_[n1] = xa_[n1] != 0.f ? xa_[n1] / ea : 0.f;
        xb_[n1] = xb_[n1] != 0.f ? xb_[n1] / eb : 0.f;
      }
      v[n1] = make_float2(xa_[n1] * w * ma, xb_[n1] * w * mb);
    }
#ifndef STABL_NO_DFT
    dft32_dif(v);
#endif
    constexpr int BREV[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                              1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
      const float2 a = v[BREV[k1]];
      const float2 w = tw[l * k1];                          // W_960^(n2 k1), n2 k1 <= 899
      zbuf[f][k1 * 31 + l] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);   // row pitch 31: conflict-free reads
    }
  }
  // pass 1 and pass 2 of an FFT run in the same half-wave: LDS executes a wave's accesses in order, no workgroup barrier needed -
  // but the wave must be converged between the writes (lanes 0..29) and the reads (lanes 0..31): a convergent no-op pins that
  // (round 3: a kernel with two `l < 30` blocks around such reads was compiled into branches that read before the other side wrote)
  __builtin_amdgcn_wave_barrier();
  {
    float2 u[30];
#pragma unroll
    for (int n2 = 0; n2 < 30; ++n2) u[n2] = zbuf[f][l * 31 + n2];
#ifndef STABL_NO_DFT
    dft30_pfa(u);
#endif
#pragma unroll
    for (int k2 = 0; k2 < 30; ++k2) zbuf[f][l + 32 * k2] = u[k2];
  }
  __syncthreads();
  int olen = T;
  if (lens != nullptr) olen = (lens[b] + N - N) / hop + 1;
  for (int idx = tid; idx < NFF * F; idx += NTH) {
    const int ff = idx / F, k = idx - ff * F;
    const float2 zk = zbuf[ff][k];
    const float2 zc = zbuf[ff][k == 0 ? 0 : N - k];
    float2 xa = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
    float2 xbv = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
    if (MODE == 1) {
      const bool edge = k == 0 || k == N / 2;
      const float sc = edge ? 1.0f / (float)N : 2.0f / (float)N;
      xa.x *= sc; xbv.x *= sc;
      xa.y = edge ? 0.f : xa.y * sc;
      xbv.y = edge ? 0.f : xbv.y * sc;
    }
    const int t = blockIdx.x * 2 * NFF + 2 * ff;
#ifdef STABL_NO_STORE
    if (xa.x != 123.456f) continue;
#endif
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    if (t < T) {
      const f32x2_t o = (t < olen) ? f32x2_t{xa.x, xa.y} : f32x2_t{0.f, 0.f};
      __builtin_nontemporal_store(o, reinterpret_cast<f32x2_t*>(out + ((size_t)b * T + t) * F + k));
    }
    if (t + 1 < T) {
      const f32x2_t o = (t + 1 < olen) ? f32x2_t{xbv.x, xbv.y} : f32x2_t{0.f, 0.f};
      __builtin_nontemporal_store(o, reinterpret_cast<f32x2_t*>(out + ((size_t)b * T + t + 1) * F + k));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 960-point STFT at hop 480, PIPELINED (round 6; replaces the "slim" experiment of round 3, which lost in the step).
// The arithmetic is stft960_kernel's (32 x 30 register FFT per half-wave, one complex transform = two real frames); what changes is
// the order in which a CU touches memory.  stft960_kernel runs load -> DFT -> LDS -> DFT -> LDS -> barrier -> store once per workgroup,
// two workgroups per CU started together: the CU's memory pipe idles while it computes and its VALUs idle while it loads (ablation,
// round 3: loads + LDS passes alone 24.9 of 36.4 us).  Here
//  (i)  a half-wave owns PP frame pairs one after the other and REQUESTS pair i + 1's samples before it transforms pair i (48 registers);
//  (ii) the half-wave splits its own complex spectrum into the two real frames' spectra and stores them: the only workgroup barrier is
//       the one behind the twiddle table, waves drift apart, one wave's loads and stores travel under another wave's DFTs;
//  (iii) samples come through a buffer descriptor over the utterance: frame a = rows 0..31, frame b = rows 16..47 of ONE 48 x 30 matrix of
//       consecutive samples (hop 480 = 16 rows), one voffset + immediates; only pairs that touch an end of the utterance take the
//       per-element path (reflect padding, MODE 0; zero padding, MODE 1).
// MODE as stft960_kernel.
template <int MODE, int PP>
__global__ void __launch_bounds__(256, 2) stft960p_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                          float2* __restrict__ out, int L, int T, int hann,
                                                          const float2* __restrict__ tw_g) {
  constexpr int N = 960, F = 481, NFF = 8, NTH = 256, HOP = 480, ZS = 1000, NR = 48;
  constexpr int OOB = 0x7ffffff0;                           // an offset every descriptor here answers with 0 (loads) / drops (stores)
  __shared__ float2 zbuf[NFF][ZS];
  __shared__ float2 tw[N];
  const int tid = threadIdx.x, l = tid & 31, f = tid >> 5;
  const int b = blockIdx.y;
  const int lq = l < 30 ? l : 29;                           // lanes 30 / 31 run pass 1 on a copy of lane 29's column and write it to spare slots
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)b * L, 0, (int)((unsigned)L * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ro =
      __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * T * F, 0, (int)((unsigned)T * F * 8u), 0x00020000);
  float s[NR];
  auto request = [&](int p) {
    const int q0 = 2 * p * HOP - N / 2;                     // position of the pair's first sample
    // wave-uniform choice (a wave's two half-waves own neighbouring pairs): the per-element form is the identity on inside positions
    if (__all(q0 >= 0 && q0 + NR * 30 <= L)) {
#ifdef STABL_NO_LOAD
      const int vo = OOB - 6000 + 0 * q0;                   // (ablation build: every load answered by the bounds check, no memory touched)
#else
      const int vo = (q0 + lq) * 4;
#endif
#pragma unroll
      for (int r = 0; r < NR; ++r) s[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo + r * 120, 0, 0));
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        int q = q0 + lq + 30 * r;
        if (MODE == 0) {                                    // reflect; pairs past the last frame are masked below, their addresses only kept inside
          q = q < 0 ? -q : q;
          q = q >= L ? 2 * (L - 1) - q : q;
        }
        // outside [0, L): the descriptor's bounds check answers 0 (= zero padding, MODE 1)
        s[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (q >= 0 && q < L) ? q * 4 : OOB, 0, 0));
      }
    }
  };
  const int p0 = blockIdx.x * (NFF * PP) + f;
  request(p0);                                              // in flight under the twiddle table's load
  for (int i = tid; i < N; i += NTH) tw[i] = tw_g[i];
  int olen = T;
  if (lens != nullptr) olen = lens[b] / HOP + 1;
  __syncthreads();
  float2* zf = zbuf[f];
#pragma unroll 1
  for (int it = 0; it < PP; ++it) {
    const int p = p0 + it * NFF, ta = 2 * p;
    const float ma = ta < T ? 1.f : 0.f, mb = ta + 1 < T ? 1.f : 0.f;
    // the window and twiddle reads do not depend on the pair: left visible, the compiler hoists them out of the loop and spills;
    // an opaque zero in the index keeps them where they are
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const float2* twl = tw + opq;
    if (MODE == 1) {
      // divide by the OLA envelope.  It belongs to the sample's padded position m = t * 480 + i (frames m / 480 with window sample r and
      // m / 480 - 1 with r + 480), not to the frame: rows 16..31 are frame a's second half AND frame b's first - one division per row
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int rr = 30 * (r & 15) + lq, j = ta + (r >> 4);
        const float w0 = hann ? 0.5f - 0.5f * twl[rr].x : 1.0f, w1 = hann ? 0.5f - 0.5f * twl[rr + 480].x : 1.0f;
        const float e = (j < T ? w0 * w0 : 0.f) + (j >= 1 ? w1 * w1 : 0.f);
        s[r] = s[r] != 0.f ? s[r] / e : 0.f;
      }
    }
    float2 v[32];
#pragma unroll
    for (int n1 = 0; n1 < 32; ++n1) {
      const float w = hann ? 0.5f - 0.5f * twl[30 * n1 + lq].x : 1.0f;
      v[n1] = make_float2(s[n1] * w * ma, s[n1 + 16] * w * mb);
    }
    if (it + 1 < PP) request(p + NFF);                      // the next pair's samples travel under this pair's transforms
#ifndef STABL_NO_DFT
    dft32_dif(v);
#endif
    {
      constexpr int BREV[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                                1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
      float2* zw = zf + (l < 30 ? l : 992 - 30 + l);        // row pitch 31; lanes 30 / 31: slots 992.. behind the 32 rows
      const int zstep = l < 30 ? 31 : 2;
#pragma unroll
      for (int k1 = 0; k1 < 32; ++k1) {
        const float2 a = v[BREV[k1]];
        const float2 w = twl[lq * k1];                      // W_960^(n2 k1), n2 k1 <= 899
        zw[(k1 & (l < 30 ? 31 : 3)) * zstep] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
      }
    }
    __builtin_amdgcn_wave_barrier();                        // LDS runs a wave's accesses in order; the wave must be converged between them (stft960_kernel)
    {
      float2 u[30];
#pragma unroll
      for (int n2 = 0; n2 < 30; ++n2) u[n2] = zf[l * 31 + n2];
#ifndef STABL_NO_DFT
      dft30_pfa(u);
#endif
      __builtin_amdgcn_wave_barrier();                      // (all rows read before natural-order bins overwrite them)
#pragma unroll
      for (int k2 = 0; k2 < 30; ++k2) zf[l + 32 * k2] = u[k2];
    }
    __builtin_amdgcn_wave_barrier();
    // split: frame a = (Z[k] + conj Z[N - k]) / 2, frame b = -i (Z[k] - conj Z[N - k]) / 2; 32 consecutive bins per instruction,
    // through a descriptor over the utterance's spectra: frames past T / the lanes past bin 480 store to an offset it drops
    const int oa = (ta < T) ? (ta * F + l) * 8 : OOB, ob = (ta + 1 < T) ? ((ta + 1) * F + l) * 8 : OOB;
    const float keep_a = ta < olen ? 1.f : 0.f, keep_b = ta + 1 < olen ? 1.f : 0.f;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = j * 32 + l;
      const int kk = (j < 15 || l == 0) ? k : 0;
      const float2 zk = zf[kk];
      const float2 zc = zf[kk == 0 ? 0 : N - kk];
      float2 xa = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
      float2 xbv = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
      if (MODE == 1) {
        const bool edge = kk == 0 || kk == N / 2;
        const float sc = edge ? 1.0f / (float)N : 2.0f / (float)N;
        xa.x *= sc; xbv.x *= sc;
        xa.y = edge ? 0.f : xa.y * sc;
        xbv.y = edge ? 0.f : xbv.y * sc;
      }
#ifdef STABL_NO_STORE
      const bool live = xa.x == 123.456f;
#else
      const bool live = j < 15 || l == 0;
#endif
      __builtin_amdgcn_raw_buffer_store_b64(f32x2_t{xa.x * keep_a, xa.y * keep_a}, ro, live ? oa + j * 256 : OOB, 0, 2);     // aux 2 = nt
      __builtin_amdgcn_raw_buffer_store_b64(f32x2_t{xbv.x * keep_b, xbv.y * keep_b}, ro, live ? ob + j * 256 : OOB, 0, 2);
    }
    __builtin_amdgcn_wave_barrier();                        // the split's reads sit in front of the next pair's pass-1 writes
  }
}

template <int MODE>
static void launch_stft960p(const float* x, const int32_t* lens, float2* out, int B, int L, int T, int hann, const float2* tw, hipStream_t st) {
  static const int pp = [] { const char* e = getenv("URSE_STFT960_PP"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 4 ? v : 2; }();
  const int npairs = (T + 1) / 2;
  const dim3 grid(ceil_div(npairs, 8 * pp), B);
  switch (pp) {
    case 1: hipLaunchKernelGGL((stft960p_kernel<MODE, 1>), grid, dim3(256), 0, st, x, lens, out, L, T, hann, tw); break;
    case 3: hipLaunchKernelGGL((stft960p_kernel<MODE, 3>), grid, dim3(256), 0, st, x, lens, out, L, T, hann, tw); break;
    case 4: hipLaunchKernelGGL((stft960p_kernel<MODE, 4>), grid, dim3(256), 0, st, x, lens, out, L, T, hann, tw); break;
    default: hipLaunchKernelGGL((stft960p_kernel<MODE, 2>), grid, dim3(256), 0, st, x, lens, out, L, T, hann, tw); break;
  }
}


// iSTFT: each workgroup owns C*hop consecutive positions of the padded OLA axis and transforms
// the 2*NF frames that cover them (halo frames are recomputed, nothing is accumulated in HBM).
__global__ void __launch_bounds__(256) istft_kernel(const float2* __restrict__ spec, float* __restrict__ out, int T,
                                                    int L_out, FftPlan plan, int hop, const float* __restrict__ win_g,
                                                    const float2* __restrict__ tw_g, int NF, int C, int ov) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = plan.n;
  float2* tw = reinterpret_cast<float2*>(smem);
  float* win = reinterpret_cast<float*>(smem + (size_t)n * 8);
  float2* bufA = reinterpret_cast<float2*>(smem + (size_t)n * 12);
  float2* bufB = bufA + (size_t)NF * n;
  const int b = blockIdx.y;
  const int half = n / 2;
  const int F = half + 1;
  const bool even = (n & 1) == 0;
  const int c0 = blockIdx.x * C;   // first chunk frame index (m0 = c0*hop)
  const int tf = c0 - (ov - 1);    // first frame carried by this workgroup
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    tw[i] = tw_g[i];
    win[i] = win_g[i];
  }
  // load conj(Xa_full + i*Xb_full)
  for (int idx = threadIdx.x; idx < NF * n; idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_n), k = idx - f * n;
    const int kk = (k <= half) ? k : n - k;
    const bool edge = (kk == 0) || (even && kk == half);
    float2 X[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int t = tf + 2 * f + s;
      float2 v = make_float2(0.f, 0.f);
      if (t >= 0 && t < T) {
        v = spec[((size_t)b * T + t) * F + kk];
        if (edge) v.y = 0.f;
        if (k > half) v.y = -v.y;
      }
      X[s] = v;
    }
    bufA[idx] = make_float2(X[0].x - X[1].y, -(X[0].y + X[1].x));
  }
  __syncthreads();
  float2* Y = fft_lds_forward(bufA, bufB, NF, plan, tw);
  float* frames = reinterpret_cast<float*>(Y == bufA ? bufB : bufA);  // [2*NF][n]
  const float inv_n = 1.0f / (float)n;
  for (int idx = threadIdx.x; idx < NF * n; idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_n), i = idx - f * n;
    const float2 y = Y[idx];
    const float w = win[i] * inv_n;
    frames[(2 * f) * n + i] = y.x * w;
    frames[(2 * f + 1) * n + i] = -y.y * w;
  }
  __syncthreads();
  const int m0 = c0 * hop;
  for (int p = threadIdx.x; p < C * hop; p += blockDim.x) {
    const int m = m0 + p;
    const int o = m - half;
    if (o < 0 || o >= L_out) continue;
    float acc = 0.f, env = 0.f;
    for (int s = 0; s < 2 * NF; ++s) {
      const int t = tf + s;
      const int i = m - t * hop;
      if (t >= 0 && t < T && i >= 0 && i < n) {
        acc += frames[s * n + i];
        const float w = win[i];
        env += w * w;
      }
    }
    out[(size_t)b * L_out + o] = (env > 1e-11f) ? acc / env : 0.f;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// 960-point inverse STFT at hop 480 (the C2 back end), on the register FFT of stft960_kernel.  A workgroup owns 15 hops of
// the padded overlap-add axis and transforms the 16 frames that cover them (one halo frame recomputed); a half-wave
// turns one frame PAIR into one complex transform: z[k] = conj(Xa[k] + i Xb[k]) over the Hermitian-extended spectra,
// y = FFT(z), frame a = Re y / n, frame b = -Im y / n.  The generic kernel above walks five radix passes through LDS
// (96 us at C2); here a transform makes two LDS round trips and the windowed overlap-add reads the complex result in place.
constexpr int IS960_NFF = 8, IS960_C = 2 * IS960_NFF - 1;
__global__ void __launch_bounds__(IS960_NFF * 32) istft960_kernel(const float2* __restrict__ spec, float* __restrict__ out, int T,
                                                                   int L_out, int hann, const float2* __restrict__ tw_g) {
  constexpr int N = 960, F = 481, HOP = 480, NFF = IS960_NFF, NTH = NFF * 32, ZS = 1000;
  __shared__ float2 zbuf[NFF][ZS];
  __shared__ float2 tw[N];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31;
  const int f = 2 * (tid >> 6) + (lane >> 5);
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * IS960_C;          // first hop of the chunk
  const int tf = c0 - 1;                        // first frame carried
  for (int i = tid; i < N; i += NTH) tw[i] = tw_g[i];
  const int ta = tf + 2 * f, tb = ta + 1;
  const bool va = ta >= 0 && ta < T, vb = tb >= 0 && tb < T;
  const float2* sa = spec + ((size_t)b * T + (va ? ta : 0)) * F;
  const float2* sb = spec + ((size_t)b * T + (vb ? tb : 0)) * F;
  const float ma = va ? 1.f : 0.f, mb = vb ? 1.f : 0.f;
  const int lq = l < 30 ? l : 29;
  // rows 0..15 (k = 30 n1 + l <= 479) and k = 480 are loaded; the Hermitian half comes from the other lanes' registers:
  // k' = 960 - (30 n1 + l) = 30 (31 - n1) + (30 - l), i.e. row 31 - n1 of lane 30 - l (lane 0: its own row 32 - n1)
  float2 da[17], db[17];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    da[n1] = sa[30 * n1 + lq];
    db[n1] = sb[30 * n1 + lq];
  }
  da[16] = sa[N / 2];
  db[16] = sb[N / 2];
  float2 v[32];
  const int src = (lane & 32) | (l == 0 ? 0 : (l < 30 ? 30 - l : 1));
#pragma unroll
  for (int n1 = 0; n1 < 32; ++n1) {
    float2 xa, xb;
    float sg = 1.f;
    if (n1 < 16) {
      xa = da[n1]; xb = db[n1];
    } else {
      const int m = 31 - n1;                                      // 15 .. 0
      float2 ma_ = make_float2(__shfl(da[m].x, src, 64), __shfl(da[m].y, src, 64));
      float2 mb_ = make_float2(__shfl(db[m].x, src, 64), __shfl(db[m].y, src, 64));
      // lane 0 of row n1: k = 30 n1 -> mirrored k' = 30 (32 - n1) is its own row 32 - n1 (row 16 = the Nyquist bin, unmirrored)
      const float2 oa = da[32 - n1], ob = db[32 - n1];
      xa = l == 0 ? oa : ma_;
      xb = l == 0 ? ob : mb_;
      sg = (n1 == 16 && l == 0) ? 1.f : -1.f;
    }
    const bool edge = (n1 == 0 || n1 == 16) && l == 0;            // DC / Nyquist: imaginary parts dropped
    const float e = edge ? 0.f : 1.f;
    const float ax = xa.x * ma, ay = xa.y * (ma * sg * e), bx = xb.x * mb, by = xb.y * (mb * sg * e);
    v[n1] = make_float2(ax - by, -(ay + bx));
  }
  __syncthreads();                                               // twiddle table complete
  if (l < 30) {
    dft32_dif(v);
    constexpr int BREV[32] = {0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30,
                              1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31};
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
      const float2 a = v[BREV[k1]];
      const float2 w = tw[l * k1];
      zbuf[f][k1 * 31 + l] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
    }
  }
  __builtin_amdgcn_wave_barrier();        // converged between the half-wave's LDS writes and reads (see stft960_kernel)
  {
    float2 u[30];
#pragma unroll
    for (int n2 = 0; n2 < 30; ++n2) u[n2] = zbuf[f][l * 31 + n2];
    dft30_pfa(u);
#pragma unroll
    for (int k2 = 0; k2 < 30; ++k2) zbuf[f][l + 32 * k2] = u[k2];
  }
  __syncthreads();
  // overlap-add: position p of the chunk takes sample 480 + p % 480 of frame s1 = p / 480 and sample p % 480 of frame s1 + 1
  const float inv_n = 1.0f / (float)N;
  for (int p = tid; p < IS960_C * HOP; p += NTH) {
    const int o = c0 * HOP + p - N / 2;
    if (o < 0 || o >= L_out) continue;
    const int s1 = p / HOP, i2 = p - s1 * HOP, i1 = i2 + HOP, s2 = s1 + 1;
    const float w1 = hann ? 0.5f - 0.5f * tw[i1].x : 1.f, w2 = hann ? 0.5f - 0.5f * tw[i2].x : 1.f;
    const float2 y1 = zbuf[s1 >> 1][i1], y2 = zbuf[s2 >> 1][i2];
    const float f1 = (s1 & 1) ? -y1.y : y1.x, f2 = (s2 & 1) ? -y2.y : y2.x;
    const bool ok1 = tf + s1 >= 0 && tf + s1 < T, ok2 = tf + s2 >= 0 && tf + s2 < T;
    const float acc = (ok1 ? f1 * (w1 * inv_n) : 0.f) + (ok2 ? f2 * (w2 * inv_n) : 0.f);
    const float env = (ok1 ? w1 * w1 : 0.f) + (ok2 ? w2 * w2 : 0.f);
    out[(size_t)b * L_out + o] = (env > 1e-11f) ? acc / env : 0.f;
  }
}

}  // namespace urse

using namespace urse;

static void init_lds_attrs() {
  std::call_once(g_lds_once, [] {
    allow_big_lds(stft_kernel<0>);
    allow_big_lds(stft_kernel<1>);
    allow_big_lds(istft_kernel);
  });
}

extern "C" int urse_stft_fwd(const float* wav, const int32_t* lens, float* spec, int B, int L, int n_fft, int hop,
                             int window, void* stream) {
  URSE_CHECK_ARG(wav && spec && B > 0 && L > 0 && hop > 0 && n_fft >= 2, "urse_stft_fwd: bad argument");
  URSE_CHECK_ARG(n_fft / 2 < L, "urse_stft_fwd: reflect padding needs n_fft/2 (%d) < L (%d)", n_fft / 2, L);
  URSE_CHECK_ARG(window == URSE_WIN_RECT || window == URSE_WIN_HANN, "urse_stft_fwd: unknown window %d", window);
  StftTables tb;
  init_lds_attrs();
  int rc = get_tables(n_fft, &tb);
  if (rc) return rc;
  const int T = L / hop + 1;
  static const bool no960 = getenv("URSE_STFT_GENERIC") != nullptr;
  if (n_fft == 960 && !no960) {
    note_launch(URSE_KV_STFT960);
    // hop 480 (the C2 front end): the pipelined kernel; URSE_STFT960_PIPE=0 keeps the round-2 kernel (A/B switch); other hops: round-2 kernel
    static const bool pipe = !(getenv("URSE_STFT960_PIPE") != nullptr && atoi(getenv("URSE_STFT960_PIPE")) == 0);
    if (pipe && hop == 480 && (long)L * 4 < (1L << 31))
      launch_stft960p<0>(wav, lens, reinterpret_cast<float2*>(spec), B, L, T, window == URSE_WIN_HANN ? 1 : 0, tb.tw, (hipStream_t)stream);
    else
    hipLaunchKernelGGL(stft960_kernel<0>, dim3(ceil_div(T, 2 * URSE_STFT960_NFF), B), dim3(URSE_STFT960_NFF_THREADS), 0, (hipStream_t)stream, wav, lens,
                       reinterpret_cast<float2*>(spec), L, T, hop, window == URSE_WIN_HANN ? 1 : 0, tb.tw);
    URSE_CHECK_LAUNCH("urse_stft_fwd");
    return URSE_OK;
  }
  const int NF = pick_nf(n_fft, 1);
  dim3 grid(ceil_div(T, 2 * NF), B);
  note_launch(URSE_KV_STFT_GENERIC);
  hipLaunchKernelGGL(stft_kernel<0>, grid, dim3(stft_threads()), lds_bytes(n_fft, NF), (hipStream_t)stream, wav, lens,
                     reinterpret_cast<float2*>(spec), L, T, tb.plan, hop, tb.win[window], tb.tw, NF);
  URSE_CHECK_LAUNCH("urse_stft_fwd");
  return URSE_OK;
}

extern "C" int urse_istft_bwd(const float* grad_wav, float* grad_spec, int B, int T, int n_fft, int hop, int L_out,
                              int window, void* stream) {
  URSE_CHECK_ARG(grad_wav && grad_spec && B > 0 && T > 0 && hop > 0 && n_fft >= 2 && L_out > 0,
                 "urse_istft_bwd: bad argument");
  StftTables tb;
  init_lds_attrs();
  int rc = get_tables(n_fft, &tb);
  if (rc) return rc;
  static const bool no960 = getenv("URSE_STFT_GENERIC") != nullptr;
  if (n_fft == 960 && hop == 480 && L_out > 480 && (window == URSE_WIN_RECT || window == URSE_WIN_HANN) && !no960) {
    note_launch(URSE_KV_STFT960);
    static const bool pipe = !(getenv("URSE_STFT960_PIPE") != nullptr && atoi(getenv("URSE_STFT960_PIPE")) == 0);
    if (pipe && (long)L_out * 4 < (1L << 31))
      launch_stft960p<1>(grad_wav, nullptr, reinterpret_cast<float2*>(grad_spec), B, L_out, T, window == URSE_WIN_HANN ? 1 : 0, tb.tw, (hipStream_t)stream);
    else
    hipLaunchKernelGGL(stft960_kernel<1>, dim3(ceil_div(T, 2 * URSE_STFT960_NFF), B), dim3(URSE_STFT960_NFF_THREADS), 0, (hipStream_t)stream,
                       grad_wav, (const int32_t*)nullptr, reinterpret_cast<float2*>(grad_spec), L_out, T, hop,
                       window == URSE_WIN_HANN ? 1 : 0, tb.tw);
    URSE_CHECK_LAUNCH("urse_istft_bwd");
    return URSE_OK;
  }
  const int NF = pick_nf(n_fft, 1);
  dim3 grid(ceil_div(T, 2 * NF), B);
  hipLaunchKernelGGL(stft_kernel<1>, grid, dim3(stft_threads()), lds_bytes(n_fft, NF), (hipStream_t)stream, grad_wav,
                     (const int32_t*)nullptr, reinterpret_cast<float2*>(grad_spec), L_out, T, tb.plan, hop,
                     tb.win[window], tb.tw, NF);
  URSE_CHECK_LAUNCH("urse_istft_bwd");
  return URSE_OK;
}

extern "C" int urse_istft_fwd(const float* spec, float* wav, int B, int T, int n_fft, int hop, int L_out, int window,
                              void* stream) {
  URSE_CHECK_ARG(spec && wav && B > 0 && T > 0 && hop > 0 && n_fft >= 2 && L_out > 0, "urse_istft_fwd: bad argument");
  StftTables tb;
  init_lds_attrs();
  int rc = get_tables(n_fft, &tb);
  if (rc) return rc;
  static const bool no960 = getenv("URSE_STFT_GENERIC") != nullptr;
  if (n_fft == 960 && hop == 480 && (window == URSE_WIN_RECT || window == URSE_WIN_HANN) && !no960) {
    const long need960 = (long)n_fft / 2 + L_out;
    note_launch(URSE_KV_ISTFT960);
    hipLaunchKernelGGL(istft960_kernel, dim3(ceil_div(need960, (long)IS960_C * 480), B), dim3(IS960_NFF * 32), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(spec), wav, T, L_out, window == URSE_WIN_HANN ? 1 : 0, tb.tw);
    URSE_CHECK_LAUNCH("urse_istft_fwd");
    return URSE_OK;
  }
  const int ov = (n_fft + hop - 1) / hop;
  const int NF = pick_nf(n_fft, (ov + 2) / 2);
  const int C = 2 * NF - ov + 1;
  URSE_CHECK_ARG(C >= 1, "urse_istft_fwd: hop %d too small for n_fft %d", hop, n_fft);
  URSE_CHECK_ARG(lds_bytes(n_fft, NF) <= 160 * 1024, "urse_istft_fwd: n_fft %d / hop %d exceeds LDS", n_fft, hop);
  // padded axis covers positions [0, n + hop*(T-1)); only [half, half + L_out) is written
  const long need = (long)n_fft / 2 + L_out;
  dim3 grid(ceil_div(need, (long)C * hop), B);
  note_launch(URSE_KV_ISTFT_GENERIC);
  hipLaunchKernelGGL(istft_kernel, grid, dim3(stft_threads()), lds_bytes(n_fft, NF), (hipStream_t)stream,
                     reinterpret_cast<const float2*>(spec), wav, T, L_out, tb.plan, hop, tb.win[window], tb.tw, NF, C,
                     ov);
  URSE_CHECK_LAUNCH("urse_istft_fwd");
  return URSE_OK;
}
