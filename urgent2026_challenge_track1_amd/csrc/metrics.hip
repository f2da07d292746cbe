// Batched intrusive metrics: ESTOI (pystoi.stoi(extended=True)) and SDR (fast_bss_eval.bss_eval_sources,
// one source, 512-tap distortion filter) for P pairs at once, one workgroup per pair where the algorithm is
// sequential (silent-frame compaction, Levinson recursion) and (chunk x pair) grids where it is not.
// Reference call sites: evaluation_metrics/calculate_intrusive_se_metrics.py:37-48 (estoi_metric), :90-109
// (sdr_metric).  HBM traffic is one read of each pair; everything else lives in L2 / LDS.  Accumulations that decide
// thresholds or near-cancelling ratios (frame energies, correlations, the Toeplitz solve) are f64.
#include <math.h>

#include "fft_lds.h"

namespace urse {

// ---- polyphase resampler == scipy.signal.resample_poly(x, up, down, window=h) as used by pystoi.resample_oct ----
// y[n] = sum_j x[j] * hp[(n + n_pre_remove) * down - j * up],  hp = zero-front-padded up*h (host-built, f64)
__global__ void __launch_bounds__(256) resample_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const double* __restrict__ hp, int hlen, int L, int Lout,
                                                       int up, int down, int n_pre_remove) {
  const int p = blockIdx.y;
  const float* xp = x + (long)p * L;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Lout; n += gridDim.x * blockDim.x) {
    const long m = (long)(n + n_pre_remove) * down;
    // need 0 <= m - j*up < hlen  ->  j in [ceil((m-hlen+1)/up), floor(m/up)]
    long jhi = m / up;
    long jlo = (m - hlen + 1 + up - 1) / up;
    if (m - hlen + 1 <= 0) jlo = 0;
    if (jhi > L - 1) jhi = L - 1;
    double acc = 0.0;
    for (long j = jlo; j <= jhi; ++j) acc += (double)xp[j] * hp[m - j * up];
    y[(long)p * Lout + n] = (float)acc;
  }
}

__device__ __forceinline__ float hann258(int i) {  // np.hanning(258)[1:-1][i]
  return 0.5f - 0.5f * cosf(6.28318530717958647692f * (float)(i + 1) / 257.0f);
}

// ---- remove_silent_frames: one workgroup per pair ------------------------------------------------------------
// dynamic LDS: double en[nfr_max] | int src[nfr_max] | int scan[nfr_max]
__global__ void __launch_bounds__(256) silent_frames_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            float* __restrict__ xs, float* __restrict__ ys,
                                                            int* __restrict__ len_out, int L, int nfr_max) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[4];
  __shared__ int nkept_s;
  double* en = reinterpret_cast<double*>(smem);
  int* src = reinterpret_cast<int*>(en + nfr_max);
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* xp = x + (long)p * L;
  const float* yp = y + (long)p * L;
  const int nfr = L > 256 ? (L - 256 + 127) / 128 : 0;   // len(range(0, L - 256, 128))
  // frame energies 20*log10(||w * x_frame|| + EPS)
  for (int f = w; f < nfr; f += 4) {
    double s = 0.0;
    for (int i = lane; i < 256; i += 64) {
      const double v = (double)hann258(i) * (double)xp[f * 128 + i];
      s += v * v;
    }
    s = wave_sum_d(s);
    if (lane == 0) en[f] = 20.0 * log10(sqrt(s) + 2.220446049250313e-16);
  }
  __syncthreads();
  double mx = -1e300;
  for (int f = tid; f < nfr; f += 256) mx = fmax(mx, en[f]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  // compaction (serial scan by one thread: nfr is a few hundred)
  if (tid == 0) {
    int k = 0;
    for (int f = 0; f < nfr; ++f)
      if ((mx - 40.0 - en[f]) < 0.0) src[k++] = f;
    nkept_s = k;
  }
  __syncthreads();
  const int nk = nkept_s;
  const int Ls = nk > 0 ? (nk - 1) * 128 + 256 : 0;
  if (tid == 0) len_out[p] = Ls;
  for (int s = tid; s < Ls; s += 256) {
    const int k1 = s >> 7;
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int k = k1 - d;
      if (k < 0 || k >= nk) continue;
      const int i = s - k * 128;
      if (i >= 256) continue;
      const float wv = hann258(i);
      const int o = src[k] * 128 + i;
      ax += wv * xp[o];
      ay += wv * yp[o];
    }
    xs[(long)p * L + s] = ax;
    ys[(long)p * L + s] = ay;
  }
}

// ---- STFT (256-sample Hann frames, 512-point FFT, hop 128) + third-octave band energies -----------------------------
struct TobBands { int lo[15]; int hi[15]; };

__global__ void __launch_bounds__(256) tob_kernel(const float* __restrict__ xs, const float* __restrict__ ys,
                                                  const int* __restrict__ lens, float* __restrict__ tobx,
                                                  float* __restrict__ toby, int L, int Mmax, FftPlan plan,
                                                  const float2* __restrict__ tw_g, TobBands bands) {
  constexpr int NF = 4, n = 512;
  __shared__ __attribute__((aligned(16))) float2 tw[n];
  __shared__ __attribute__((aligned(16))) float2 bufA[NF * n];
  __shared__ __attribute__((aligned(16))) float2 bufB[NF * n];
  const int p = blockIdx.y, m0 = blockIdx.x * NF, tid = threadIdx.x;
  const int Ls = lens[p];
  const int M = Ls > 256 ? (Ls - 256 + 127) / 128 : 0;
  if (m0 >= M) return;
  for (int i = tid; i < n; i += 256) tw[i] = tw_g[i];
  const float* xp = xs + (long)p * L;
  const float* yp = ys + (long)p * L;
  for (int idx = tid; idx < NF * n; idx += 256) {
    const int f = idx >> 9, i = idx & 511;
    float2 v = make_float2(0.f, 0.f);
    if (i < 256 && m0 + f < M) {
      const float wv = hann258(i);
      const int o = (m0 + f) * 128 + i;
      v = make_float2(wv * xp[o], wv * yp[o]);
    }
    bufA[idx] = v;
  }
  __syncthreads();
  const float2* Z = fft_lds_forward(bufA, bufB, NF, plan, tw);
  // 15 bands x NF frames x 2 signals: one thread per (frame, band)
  for (int idx = tid; idx < NF * 15; idx += 256) {
    const int f = idx / 15, b = idx - f * 15;
    if (m0 + f >= M) continue;
    float sx = 0.f, sy = 0.f;
    for (int k = bands.lo[b]; k < bands.hi[b]; ++k) {
      const float2 zk = Z[f * n + k], zc = Z[f * n + (k == 0 ? 0 : n - k)];
      const float xr = 0.5f * (zk.x + zc.x), xi = 0.5f * (zk.y - zc.y);
      const float yr = 0.5f * (zk.y + zc.y), yi = -0.5f * (zk.x - zc.x);
      sx += xr * xr + xi * xi;
      sy += yr * yr + yi * yi;
    }
    tobx[((long)p * 15 + b) * Mmax + m0 + f] = sqrtf(sx);
    toby[((long)p * 15 + b) * Mmax + m0 + f] = sqrtf(sy);
  }
}

// ---- ESTOI: 30-frame segments, row then column normalisation, mean correlation; one workgroup per pair ---------
__global__ void __launch_bounds__(256) estoi_kernel(const float* __restrict__ tobx, const float* __restrict__ toby,
                                                    const int* __restrict__ lens, float* __restrict__ out, int Mmax) {
  __shared__ double segx[4][15][30], segy[4][15][30];
  __shared__ double red[4];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Ls = lens[p];
  const int M = Ls > 256 ? (Ls - 256 + 127) / 128 : 0;
  if (M < 30) {
    if (tid == 0) out[p] = 1e-5f;
    return;
  }
  const int J = M - 29;
  const float* tx = tobx + (long)p * 15 * Mmax;
  const float* ty = toby + (long)p * 15 * Mmax;
  double total = 0.0;
  for (int j = w; j < J; j += 4) {
    for (int e = lane; e < 450; e += 64) {
      const int b = e / 30, c = e - b * 30;
      segx[w][b][c] = (double)tx[b * Mmax + j + c];
      segy[w][b][c] = (double)ty[b * Mmax + j + c];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 30) {  // rows: lanes 0-14 -> x band rows, 15-29 -> y band rows
      double(*sg)[30] = lane < 15 ? segx[w] : segy[w];
      const int b = lane < 15 ? lane : lane - 15;
      double mu = 0.0;
      for (int c = 0; c < 30; ++c) mu += sg[b][c];
      mu /= 30.0;
      double ss = 0.0;
      for (int c = 0; c < 30; ++c) { const double v = sg[b][c] - mu; ss += v * v; }
      const double inv = 1.0 / sqrt(ss);
      for (int c = 0; c < 30; ++c) sg[b][c] = (sg[b][c] - mu) * inv;
    }
    __builtin_amdgcn_wave_barrier();
    double part = 0.0;
    if (lane < 30) {  // columns: lane c normalises column c of both and correlates
      const int c = lane;
      double mx = 0.0, my = 0.0;
      for (int b = 0; b < 15; ++b) { mx += segx[w][b][c]; my += segy[w][b][c]; }
      mx /= 15.0; my /= 15.0;
      double sx = 0.0, sy = 0.0, sxy = 0.0;
      for (int b = 0; b < 15; ++b) {
        const double vx = segx[w][b][c] - mx, vy = segy[w][b][c] - my;
        sx += vx * vx; sy += vy * vy; sxy += vx * vy;
      }
      part = sxy / (sqrt(sx) * sqrt(sy));
    }
    total += part;
    __builtin_amdgcn_wave_barrier();
  }
  total = wave_sum_d(total);
  if (lane == 0) red[w] = total;
  __syncthreads();
  if (tid == 0) out[p] = (float)((red[0] + red[1] + red[2] + red[3]) / 30.0 / (double)J);
}

// ---- SDR: auto / cross correlations for 512 lags in f64 (time domain, LDS-tiled) -------------------------------
constexpr int XC_CHUNK = 2048, XC_LAGS = 512;
__global__ void __launch_bounds__(256) xcorr_kernel(const float* __restrict__ ref, const float* __restrict__ est,
                                                    double* __restrict__ acf, double* __restrict__ xc,
                                                    double* __restrict__ norms, int L) {
  __shared__ double r[XC_CHUNK + XC_LAGS], e[XC_CHUNK + XC_LAGS];
  __shared__ double red[2][4];
  const int p = blockIdx.y, n0 = blockIdx.x * XC_CHUNK, tid = threadIdx.x;
  const float* rp = ref + (long)p * L;
  const float* ep = est + (long)p * L;
  double nr = 0.0, ne = 0.0;
  for (int i = tid; i < XC_CHUNK + XC_LAGS; i += 256) {
    const int n = n0 + i;
    const double rv = n < L ? (double)rp[n] : 0.0, ev = n < L ? (double)ep[n] : 0.0;
    r[i] = rv;
    e[i] = ev;
    if (i < XC_CHUNK) { nr += rv * rv; ne += ev * ev; }
  }
  nr = wave_sum_d(nr);
  ne = wave_sum_d(ne);
  if ((tid & 63) == 0) { red[0][tid >> 6] = nr; red[1][tid >> 6] = ne; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(norms + (long)p * 2, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(norms + (long)p * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
  int cnt = L - n0;
  if (cnt > XC_CHUNK) cnt = XC_CHUNK;
  double a0 = 0.0, a1 = 0.0, x0 = 0.0, x1 = 0.0;   // lags tid and tid + 256
  for (int n = 0; n < cnt; ++n) {
    const double rv = r[n];
    a0 += rv * r[n + tid];
    a1 += rv * r[n + tid + 256];
    x0 += rv * e[n + tid];
    x1 += rv * e[n + tid + 256];
  }
  atomicAdd(acf + (long)p * XC_LAGS + tid, a0);
  atomicAdd(acf + (long)p * XC_LAGS + tid + 256, a1);
  atomicAdd(xc + (long)p * XC_LAGS + tid, x0);
  atomicAdd(xc + (long)p * XC_LAGS + tid + 256, x1);
}

// ---- SDR: symmetric Toeplitz solve (Levinson recursion, f64) + coherence -> dB; one workgroup (512 thr) per pair ----
__global__ void __launch_bounds__(512) sdr_solve_kernel(const double* __restrict__ acf, const double* __restrict__ xc,
                                                        const double* __restrict__ norms, float* __restrict__ out,
                                                        double clamp_eps) {
  constexpr int NL = XC_LAGS;
  __shared__ double t[NL], b[NL], f[NL], x[NL], fnew[NL];
  __shared__ double red[2][8];
  const int p = blockIdx.x, i = threadIdx.x, lane = i & 63, w = i >> 6;
  const double nr = fmax(sqrt(norms[(long)p * 2]), 1e-6), ne = fmax(sqrt(norms[(long)p * 2 + 1]), 1e-6);
  t[i] = acf[(long)p * NL + i] / (nr * nr);
  b[i] = xc[(long)p * NL + i] / (nr * ne);
  f[i] = 0.0;
  x[i] = 0.0;
  __syncthreads();
  if (i == 0) { f[0] = 1.0 / t[0]; x[0] = b[0] / t[0]; }
  __syncthreads();
  for (int k = 1; k < NL; ++k) {
    // ef = sum_{j<k} t[k-j] f[j] ; ex = sum_{j<k} t[k-j] x[j]
    double ef = 0.0, ex = 0.0;
    if (i < k) { ef = t[k - i] * f[i]; ex = t[k - i] * x[i]; }
    ef = wave_sum_d(ef);
    ex = wave_sum_d(ex);
    if (lane == 0) { red[0][w] = ef; red[1][w] = ex; }
    __syncthreads();
    ef = 0.0; ex = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { ef += red[0][q]; ex += red[1][q]; }
    const double den = 1.0 / (1.0 - ef * ef);
    // f_new = den * [f, 0] - ef * den * [0, reverse(f)] ;  backward vector = reverse(f_new)
    if (i <= k) {
      const double fi = i < k ? f[i] : 0.0;
      const double bi = i >= 1 ? f[k - i] : 0.0;   // [0, reverse(f)][i] = f[k-1-(i-1)]
      fnew[i] = den * fi - ef * den * bi;
    }
    __syncthreads();
    if (i <= k) {
      f[i] = fnew[i];
      x[i] = (i < k ? x[i] : 0.0) + (b[k] - ex) * fnew[k - i];   // + (b_k - ex) * backward_new[i]
    }
    __syncthreads();
  }
  double c = b[i] * x[i];
  c = wave_sum_d(c);
  if (lane == 0) red[0][w] = c;
  __syncthreads();
  if (i == 0) {
    double coh = 0.0;
    for (int q = 0; q < 8; ++q) coh += red[0][q];
    coh = fmin(fmax(coh, clamp_eps), 1.0 - clamp_eps);
    out[p] = (float)(10.0 * log10(coh / (1.0 - coh)));
  }
}

}  // namespace urse

using namespace urse;

extern "C" int urse_resample_poly(const float* x, float* y, const double* h_padded, int hlen, int P, int L, int Lout,
                                  int up, int down, int n_pre_remove, void* stream) {
  URSE_CHECK_ARG(x && y && h_padded && hlen > 0 && P > 0 && L > 0 && Lout > 0 && up > 0 && down > 0,
                 "urse_resample_poly: bad argument");
  hipLaunchKernelGGL(resample_kernel, dim3(ceil_div(Lout, 256 * 4), P), dim3(256), 0, (hipStream_t)stream, x, y,
                     h_padded, hlen, L, Lout, up, down, n_pre_remove);
  URSE_CHECK_LAUNCH("urse_resample_poly");
  return URSE_OK;
}

extern "C" int urse_estoi_batch(const float* ref10k, const float* inf10k, float* out, float* ws_x, float* ws_y,
                                float* tob_x, float* tob_y, int32_t* lens, const float* tw512, int P, int L,
                                void* stream) {
  URSE_CHECK_ARG(ref10k && inf10k && out && ws_x && ws_y && tob_x && tob_y && lens && tw512 && P > 0 && L > 0,
                 "urse_estoi_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nfr = L > 256 ? (L - 256 + 127) / 128 : 0;
  const size_t lds = (size_t)(nfr + 1) * (sizeof(double) + sizeof(int));
  URSE_CHECK_ARG(lds <= 60 * 1024, "urse_estoi_batch: signal too long (%d samples @10 kHz)", L);
  hipLaunchKernelGGL(silent_frames_kernel, dim3(P), dim3(256), lds, st, ref10k, inf10k, ws_x, ws_y, lens, L, nfr + 1);
  // third-octave band edges: thirdoct(10000, 512, 15, 150), nearest-bin rule of pystoi
  TobBands tb;
  for (int k = 0; k < 15; ++k) {
    const double fl = 150.0 * pow(2.0, (2.0 * k - 1.0) / 6.0), fh = 150.0 * pow(2.0, (2.0 * k + 1.0) / 6.0);
    int bl = 0, bh = 0;
    double dl = 1e300, dh = 1e300;
    for (int i = 0; i <= 256; ++i) {
      const double f = 10000.0 * i / 512.0;
      if ((f - fl) * (f - fl) < dl) { dl = (f - fl) * (f - fl); bl = i; }
      if ((f - fh) * (f - fh) < dh) { dh = (f - fh) * (f - fh); bh = i; }
    }
    tb.lo[k] = bl;
    tb.hi[k] = bh;
  }
  FftPlan plan;
  make_fft_plan(512, &plan);
  const int Mmax = nfr > 0 ? nfr : 1;
  hipLaunchKernelGGL(tob_kernel, dim3(ceil_div(Mmax, 4), P), dim3(256), 0, st, ws_x, ws_y, lens, tob_x, tob_y, L, Mmax,
                     plan, reinterpret_cast<const float2*>(tw512), tb);
  hipLaunchKernelGGL(estoi_kernel, dim3(P), dim3(256), 0, st, tob_x, tob_y, lens, out, Mmax);
  URSE_CHECK_LAUNCH("urse_estoi_batch");
  return URSE_OK;
}

extern "C" int urse_sdr_batch(const float* ref, const float* est, float* out, double* acf, double* xcorr,
                              double* norms, int P, int L, float clamp_db, void* stream) {
  URSE_CHECK_ARG(ref && est && out && acf && xcorr && norms && P > 0 && L > 0, "urse_sdr_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(acf, 0, sizeof(double) * XC_LAGS * P, st);
  (void)hipMemsetAsync(xcorr, 0, sizeof(double) * XC_LAGS * P, st);
  (void)hipMemsetAsync(norms, 0, sizeof(double) * 2 * P, st);
  hipLaunchKernelGGL(xcorr_kernel, dim3(ceil_div(L, XC_CHUNK), P), dim3(256), 0, st, ref, est, acf, xcorr, norms, L);
  const double e = pow(10.0, -(double)clamp_db / 10.0);
  hipLaunchKernelGGL(sdr_solve_kernel, dim3(P), dim3(512), 0, st, acf, xcorr, norms, out, e / (1.0 + e));
  URSE_CHECK_LAUNCH("urse_sdr_batch");
  return URSE_OK;
}
