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
  const int NF = pick_nf(n_fft, 1);
  dim3 grid(ceil_div(T, 2 * NF), B);
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
  const int ov = (n_fft + hop - 1) / hop;
  const int NF = pick_nf(n_fft, (ov + 2) / 2);
  const int C = 2 * NF - ov + 1;
  URSE_CHECK_ARG(C >= 1, "urse_istft_fwd: hop %d too small for n_fft %d", hop, n_fft);
  URSE_CHECK_ARG(lds_bytes(n_fft, NF) <= 160 * 1024, "urse_istft_fwd: n_fft %d / hop %d exceeds LDS", n_fft, hop);
  // padded axis covers positions [0, n + hop*(T-1)); only [half, half + L_out) is written
  const long need = (long)n_fft / 2 + L_out;
  dim3 grid(ceil_div(need, (long)C * hop), B);
  hipLaunchKernelGGL(istft_kernel, grid, dim3(stft_threads()), lds_bytes(n_fft, NF), (hipStream_t)stream,
                     reinterpret_cast<const float2*>(spec), wav, T, L_out, tb.plan, hop, tb.win[window], tb.tw, NF, C,
                     ov);
  URSE_CHECK_LAUNCH("urse_istft_fwd");
  return URSE_OK;
}
