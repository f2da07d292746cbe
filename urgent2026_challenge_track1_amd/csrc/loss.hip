// Training losses and the fused optimizer step.
//
//  * MultiResL1SpecLoss (espnet2, ctor baseline_code/d_model.py:24, call :74): unbiased-std
//    normalisation, alpha = <e,t>/(<e,e>+eps), L1 in time + mean over 4 boxcar-window STFT magnitude
//    L1 terms ("sum" reduction).  With a = alpha/sigma_e and b = 1/sigma_t the loss is
//    0.5*sum|a e - b t| + 0.125*sum_w sum | |STFT_w(a e)| - |STFT_w(b t)| |.
//    The spectral kernel packs (a*e_frame) + i*(b*t_frame) into ONE complex FFT per frame, so no
//    spectrogram ever reaches HBM; when gradients are wanted the same launch also produces
//    G = dLoss/d(a e) (magnitude-sign * phase, inverse transform, overlap-add with f32 atomics folded
//    through the reflect padding).  The backward then only applies the chain rule through a(e).
//  * SISNRLoss (d_model.py:25,80; fast_bss_eval.si_sdr_loss): closed form from the same five sums.
//  * clip_grad_norm_(0.5) + AdamW (train_se.py:78, d_model.py:104-109) on the flat buffers.
#include <map>
#include <mutex>
#include <vector>
#include <math.h>

#include "fft_lds.h"
#include "fft_reg.h"

namespace urse {

struct LossTables { FftPlan plan; float2* tw; };
static std::mutex g_lmu;
static std::map<std::pair<int, int>, LossTables> g_ltab;

static int loss_tables(int n, LossTables* out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_lmu);
  auto it = g_ltab.find({dev, n});
  if (it != g_ltab.end()) { *out = it->second; return URSE_OK; }
  LossTables t;
  if (!make_fft_plan(n, &t.plan)) { set_error("mrl1: unsupported window size %d", n); return URSE_ERR_UNSUPPORTED; }
  std::vector<float2> tw(n);
  for (int j = 0; j < n; ++j) {
    const double a = -2.0 * M_PI * (double)j / (double)n;
    tw[j] = make_float2((float)cos(a), (float)sin(a));
  }
  if (hipMalloc(&t.tw, n * sizeof(float2)) != hipSuccess) { set_error("mrl1: hipMalloc failed"); return URSE_ERR_RUNTIME; }
  (void)hipMemcpy(t.tw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice);
  g_ltab[{dev, n}] = t;
  *out = t;
  return URSE_OK;
}

// sums[b][0..4] = sum t, sum t^2, sum e, sum e^2, sum e*t   (f64)
__global__ void __launch_bounds__(256) pair_sums_kernel(const float* __restrict__ t, const float* __restrict__ e,
                                                        double* __restrict__ sums, int L, int chunk) {
  __shared__ double red[5][4];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * chunk;
  int i1 = i0 + chunk;
  if (i1 > L) i1 = L;
  const float* tb = t + (long)b * L;
  const float* eb = e + (long)b * L;
  double s[5] = {0, 0, 0, 0, 0};
  if ((L & 3) == 0 && (i0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(e)) & 15) == 0) {
    // 16-byte loads, two of each signal in flight per thread (the scalar loop ran at 1.6 TB/s: one 4-byte load per thread and trip)
    const float4* t4 = reinterpret_cast<const float4*>(tb + i0);
    const float4* e4 = reinterpret_cast<const float4*>(eb + i0);
    const int n4 = (i1 - i0) >> 2;
    auto add = [&](const float4& a, const float4& c) {
      const double tv[4] = {a.x, a.y, a.z, a.w}, ev[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) { s[0] += tv[q]; s[1] += tv[q] * tv[q]; s[2] += ev[q]; s[3] += ev[q] * ev[q]; s[4] += ev[q] * tv[q]; }
    };
    int v = threadIdx.x;
    for (; v + (int)blockDim.x < n4; v += 2 * blockDim.x) {
      const float4 a0 = t4[v], c0 = e4[v], a1 = t4[v + blockDim.x], c1 = e4[v + blockDim.x];
      add(a0, c0); add(a1, c1);
    }
    if (v < n4) add(t4[v], e4[v]);
    for (int i = i0 + (n4 << 2) + threadIdx.x; i < i1; i += blockDim.x) {
      const double tv = tb[i], ev = eb[i];
      s[0] += tv; s[1] += tv * tv; s[2] += ev; s[3] += ev * ev; s[4] += ev * tv;
    }
  } else {
    for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const double tv = tb[i], ev = eb[i];
      s[0] += tv; s[1] += tv * tv; s[2] += ev; s[3] += ev * ev; s[4] += ev * tv;
    }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    s[j] = wave_sum_d(s[j]);
    if (lane == 0) red[j][w] = s[j];
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const int j = threadIdx.x;
    atomicAdd(sums + (long)b * 5 + j, red[j][0] + red[j][1] + red[j][2] + red[j][3]);
  }
}

struct Mrl1Scalars { float a, b; double sig_t, D, mean_e, set; };

__device__ __forceinline__ Mrl1Scalars mrl1_scalars(const double* s, int L, double eps) {
  Mrl1Scalars r;
  const double var_t = (s[1] - s[0] * s[0] / L) / (L - 1);
  const double var_e = (s[3] - s[2] * s[2] / L) / (L - 1);
  r.sig_t = sqrt(var_t);
  r.D = s[3] + eps * var_e;          // a = <e,t> / (sigma_t * D)
  r.set = s[4];
  r.mean_e = s[2] / L;
  r.a = (float)(s[4] / (r.sig_t * r.D));
  r.b = (float)(1.0 / r.sig_t);
  return r;
}

// time-domain term; also initialises G = 0.5 * sign(a e - b t)
__global__ void __launch_bounds__(256) mrl1_td_kernel(const float* __restrict__ t, const float* __restrict__ e,
                                                      const double* __restrict__ sums, double* __restrict__ acc,
                                                      float* __restrict__ G, int L, int chunk, double eps, float w_td) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const Mrl1Scalars sc = mrl1_scalars(sums + (long)b * 5, L, eps);
  const int i0 = blockIdx.x * chunk;
  int i1 = i0 + chunk;
  if (i1 > L) i1 = L;
  double s = 0;
  if ((L & 3) == 0 && (i0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(G)) & 15) == 0) {
    const float4* t4 = reinterpret_cast<const float4*>(t + (long)b * L + i0);
    const float4* e4 = reinterpret_cast<const float4*>(e + (long)b * L + i0);
    float4* G4 = G ? reinterpret_cast<float4*>(G + (long)b * L + i0) : nullptr;
    const int n4 = (i1 - i0) >> 2;
    for (int v = threadIdx.x; v < n4; v += blockDim.x) {
      const float4 a = t4[v], c = e4[v];
      const float d[4] = {sc.a * c.x - sc.b * a.x, sc.a * c.y - sc.b * a.y, sc.a * c.z - sc.b * a.z, sc.a * c.w - sc.b * a.w};
      float sg[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s += fabsf(d[q]);
        sg[q] = w_td * (d[q] > 0.f ? 1.f : (d[q] < 0.f ? -1.f : 0.f));
      }
      if (G4) G4[v] = make_float4(sg[0], sg[1], sg[2], sg[3]);
    }
    for (int i = i0 + (n4 << 2) + threadIdx.x; i < i1; i += blockDim.x) {
      const float d = sc.a * e[(long)b * L + i] - sc.b * t[(long)b * L + i];
      s += fabsf(d);
      if (G) G[(long)b * L + i] = w_td * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
  } else
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const float d = sc.a * e[(long)b * L + i] - sc.b * t[(long)b * L + i];
    s += fabsf(d);
    if (G) G[(long)b * L + i] = w_td * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (long)b * 2, red[0] + red[1] + red[2] + red[3]);
}

__device__ __forceinline__ int reflect_idx(int p, int L) {
  if (p < 0) p = -p;
  if (p >= L) p = 2 * (L - 1) - p;
  return p;
}

// one boxcar-window resolution: NF frames per workgroup, frame f: z = a*e + i*b*t -> one complex FFT
template <bool GRAD>
__global__ void __launch_bounds__(256) mrl1_spec_kernel(const float* __restrict__ t, const float* __restrict__ e,
                                                        const double* __restrict__ sums, double* __restrict__ acc,
                                                        float* __restrict__ G, int L, int T, FftPlan plan, int hop,
                                                        const float2* __restrict__ tw_g, int NF, double eps,
                                                        float w_spec) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[4];
  const int n = plan.n, half = n / 2;
  float2* tw = reinterpret_cast<float2*>(smem);
  float2* bufA = tw + n;
  float2* bufB = bufA + (size_t)NF * n;
  const int b = blockIdx.y, t0 = blockIdx.x * NF;
  const Mrl1Scalars sc = mrl1_scalars(sums + (long)b * 5, L, eps);
  const float* tb = t + (long)b * L;
  const float* eb = e + (long)b * L;
  for (int i = threadIdx.x; i < n; i += blockDim.x) tw[i] = tw_g[i];
  for (int idx = threadIdx.x; idx < NF * n; idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_n), i = idx - f * n;
    const int fr = t0 + f;
    float2 v = make_float2(0.f, 0.f);
    if (fr < T) {
      const int p = reflect_idx(fr * hop + i - half, L);
      v = make_float2(sc.a * eb[p], sc.b * tb[p]);
    }
    bufA[idx] = v;
  }
  __syncthreads();
#ifdef MABL_NO_FFT
  float2* Z = bufA;
#else
  float2* Z = fft_lds_forward(bufA, bufB, NF, plan, tw);
#endif
  float2* Y = (Z == bufA) ? bufB : bufA;
  double s = 0;
  for (int idx = threadIdx.x; idx < NF * (half + 1); idx += blockDim.x) {
    const int f = fastdiv(idx, plan.m_f), k = idx - f * (half + 1);
    const float2 zk = Z[f * n + k];
    const float2 zc = Z[f * n + (k == 0 ? 0 : n - k)];
    const float2 U = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
    const float2 Tt = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
    const float mu = sqrtf(U.x * U.x + U.y * U.y), mt = sqrtf(Tt.x * Tt.x + Tt.y * Tt.y);
    const float d = mu - mt;
    if (t0 + f < T) s += fabsf(d);
    if (GRAD) {
      const float sg = (d > 0.f ? w_spec : (d < 0.f ? -w_spec : 0.f));
      float2 g = make_float2(0.f, 0.f);
      if (mu > 0.f && t0 + f < T) g = make_float2(sg * U.x / mu, sg * U.y / mu);
      const bool edge = (k == 0) || (k == half);
      if (edge) {
        Y[f * n + k] = make_float2(g.x, 0.f);
      } else {
        Y[f * n + k] = make_float2(0.5f * g.x, 0.5f * g.y);
        Y[f * n + n - k] = make_float2(0.5f * g.x, -0.5f * g.y);
      }
    }
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc + (long)b * 2 + 1, red[0] + red[1] + red[2] + red[3]);
  if (GRAD) {
    // pair frames (2p, 2p+1): in = conj(Ya + i Yb)
    const int NP = NF / 2;
    for (int idx = threadIdx.x; idx < NP * n; idx += blockDim.x) {
      const int pidx = fastdiv(idx, plan.m_n), k = idx - pidx * n;
      const float2 ya = Y[(2 * pidx) * n + k], yb = Y[(2 * pidx + 1) * n + k];
      Z[idx] = make_float2(ya.x - yb.y, -(ya.y + yb.x));
    }
    __syncthreads();
#ifdef MABL_NO_FFT
    const float2* R = Z;
#else
    const float2* R = fft_lds_forward(Z, Y, NP, plan, tw);
#endif
    for (int idx = threadIdx.x; idx < NP * n; idx += blockDim.x) {
      const int pidx = fastdiv(idx, plan.m_n), i = idx - pidx * n;
      const float2 r = R[idx];
      const int fa = t0 + 2 * pidx;
#ifdef MABL_NO_ATOMIC     // timing diagnostics (wrong results): scripts/abl_mrl1.py
      if (fa < T) G[(long)b * L + reflect_idx(fa * hop + i - half, L)] = r.x;
      if (fa + 1 < T) G[(long)b * L + reflect_idx((fa + 1) * hop + i - half, L)] = -r.y;
#elif defined(MABL_NO_SCATTER)
      if (r.x == 123.456f) G[0] = r.y;
#else
      if (fa < T) atomicAdd(G + (long)b * L + reflect_idx(fa * hop + i - half, L), r.x);
      if (fa + 1 < T) atomicAdd(G + (long)b * L + reflect_idx((fa + 1) * hop + i - half, L), -r.y);
#endif
    }
  }
}

// The same resolution on the REGISTER FFT (windows 32 * M, M = 8 / 16 / 24 / 32: the loss's 256 / 512 / 768 / 1024).  The kernel above walks
// log4(n) radix passes through LDS with run-time index arithmetic; its ablation (scripts/abl_mrl1.py, n = 1024: 162 us per launch) prices the
// FFTs at 78 us and the passes around them (frame fill, magnitudes, pair packing: one LDS trip and a barrier each) at 68.  Here a HALF-WAVE owns
// a frame from the samples to the gradient spectrum, as stft960_kernel does for the 960-point STFT:
//   pass 1: lane j (M lanes) loads z[j + M m] = a e + i b t, m = 0 .. 31 (lanes read consecutive samples), 32-point DFT in registers,
//           times W_n^(j k1), written transposed to the half-wave's LDS buffer;
//   pass 2: lane k1 (32 lanes) reads its M values, M-point DFT in registers -> Z[k1 + 32 k2], written back in natural order;
//   magnitudes / L1 / gradient spectrum in place (a lane owns the pair k, n - k);
//   gradient: the first half-wave of each wave transforms the PAIR's packed spectrum conj(Ya + i Yb) the same way and scatters
//           Re -> frame a, -Im -> frame b into G straight from its registers (lanes hold consecutive samples).
// Two LDS round trips per transform, no workgroup barrier after the twiddle table is in, every index a compile-time constant.
template <int R1, int R2, bool GRAD>
__global__ void __launch_bounds__(256) mrl1_spec_reg_kernel(const float* __restrict__ t, const float* __restrict__ e,
                                                            const double* __restrict__ sums, double* __restrict__ acc,
                                                            float* __restrict__ G, int L, int T, const float2* __restrict__ tw_g,
                                                            double eps, float w_spec) {
  // n = R1 * R2: pass 1 = R2 lanes x R1-point DFTs, pass 2 = R1 lanes x R2-point DFTs, LPF = max lanes of a frame's group
  constexpr int N = R1 * R2, HALF = N / 2, HOP = N / 2, LPF = R1 > R2 ? R1 : R2, NFR = 256 / LPF, ZP = R2 + 1, ZS = N + R1, LG1 = fr_log2(R1);
  static_assert(LPF == 16 || LPF == 32, "a frame's lane group is a quarter or a half wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double red[4];
  float2* tw = reinterpret_cast<float2*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, l = tid % LPF, hw = tid / LPF;
  float2* zb = tw + N + hw * ZS;
  const int b = blockIdx.y, fr = blockIdx.x * NFR + hw;
  const bool fok = fr < T;
  const Mrl1Scalars sc = mrl1_scalars(sums + (long)b * 5, L, eps);
  const float* tb = t + (long)b * L;
  const float* eb = e + (long)b * L;
  for (int i = tid; i < N; i += 256) tw[i] = tw_g[i];
  float2 v[R1];
  if (l < R2) {
    const int base = (fok ? fr : T - 1) * HOP - HALF + l;
    const float ma = fok ? sc.a : 0.f, mb = fok ? sc.b : 0.f;
#pragma unroll
    for (int m = 0; m < R1; ++m) {
      const int p = reflect_idx(base + R2 * m, L);
      v[m] = make_float2(ma * eb[p], mb * tb[p]);
    }
  }
  __syncthreads();                                            // the twiddle table is complete
  if (l < R2) {
#ifndef MABL_NO_FFT
    dft_pow2_dif<R1>(v);
#endif
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      const float2 a = v[fr_brev(k1, LG1)];
      const float2 w = tw[l * k1];
      zb[k1 * ZP + l] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
    }
  }
  __builtin_amdgcn_wave_barrier();        // a group's reads follow its writes in the wave's in-order LDS queue; this keeps the wave converged
  if (l < R1) {
    float2 u[R2];
#pragma unroll
    for (int j = 0; j < R2; ++j) u[j] = zb[l * ZP + j];
#ifndef MABL_NO_FFT
    dft_small<R2>(u);
#endif
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) zb[l + R1 * k2] = u[k2];
  }
  __builtin_amdgcn_wave_barrier();
  double s = 0;
#pragma unroll 1
  for (int k = l; k <= HALF; k += LPF) {
    const float2 zk = zb[k];
    const float2 zc = zb[k == 0 ? 0 : N - k];
    const float2 U = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
    const float2 Tt = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
    const float mu = sqrtf(U.x * U.x + U.y * U.y), mt = sqrtf(Tt.x * Tt.x + Tt.y * Tt.y);
    const float d = mu - mt;
    if (fok) s += fabsf(d);
    if (GRAD) {
      const float sg = (d > 0.f ? w_spec : (d < 0.f ? -w_spec : 0.f));
      float2 g = make_float2(0.f, 0.f);
      if (mu > 0.f && fok) g = make_float2(sg * U.x / mu, sg * U.y / mu);
      if (k == 0 || k == HALF) {
        zb[k] = make_float2(g.x, 0.f);
      } else {
        zb[k] = make_float2(0.5f * g.x, 0.5f * g.y);
        zb[N - k] = make_float2(0.5f * g.x, -0.5f * g.y);
      }
    }
  }
  s = wave_sum_d(s);
  if (lane == 0) red[tid >> 6] = s;
  if (GRAD) {
    __builtin_amdgcn_wave_barrier();
    // the pair (frames 2p, 2p + 1 = two adjacent lane groups of one wave): in = conj(Ya + i Yb), one transform, by the pair's first group
    if ((hw & 1) == 0) {
      const float2* za = zb;
      const float2* zo = zb + ZS;
      if (l < R2) {
#pragma unroll
        for (int m = 0; m < R1; ++m) {
          const float2 ya = za[l + R2 * m], yb = zo[l + R2 * m];
          v[m] = make_float2(ya.x - yb.y, -(ya.y + yb.x));
        }
#ifndef MABL_NO_FFT
        dft_pow2_dif<R1>(v);
#endif
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
          const float2 a = v[fr_brev(k1, LG1)];
          const float2 w = tw[l * k1];
          zb[k1 * ZP + l] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (l < R1) {
        float2 u[R2];
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = zb[l * ZP + j];
#ifndef MABL_NO_FFT
        dft_small<R2>(u);
#endif
        // y = FFT(in): Re -> frame a (samples pa0 + i), -Im -> frame b (one hop later).  A lane holds i = l + R1 k2, and one hop is R2 / 2
        // values of k2: the two frames' overlap is summed in registers, so the pair issues three half-frames of atomics instead of four
        float* Gb = G + (long)b * L;
        const int pa0 = fr * HOP - HALF + l;
        const bool oka = fr < T, okb = fr + 1 < T;
#ifndef MABL_NO_SCATTER
#pragma unroll
        for (int k2 = 0; k2 < R2 / 2; ++k2)
          if (oka) atomicAdd(Gb + reflect_idx(pa0 + R1 * k2, L), u[k2].x);
#pragma unroll
        for (int k2 = 0; k2 < R2 / 2; ++k2) {
          const float ov = (oka ? u[k2 + R2 / 2].x : 0.f) - (okb ? u[k2].y : 0.f);
          if (oka) atomicAdd(Gb + reflect_idx(pa0 + HOP + R1 * k2, L), ov);
        }
#pragma unroll
        for (int k2 = R2 / 2; k2 < R2; ++k2)
          if (okb) atomicAdd(Gb + reflect_idx(pa0 + HOP + R1 * k2, L), -u[k2].y);
#else
        if (u[0].x == 123.456f) Gb[0] = u[1].y;
#endif
      }
    }
  }
  __syncthreads();
  if (tid == 0) atomicAdd(acc + (long)b * 2 + 1, red[0] + red[1] + red[2] + red[3]);
}

__global__ void mrl1_final_kernel(const double* __restrict__ acc, float* __restrict__ loss, int B, float w_td,
                                  float w_spec) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) loss[b] = (float)(w_td * acc[b * 2] + w_spec * acc[b * 2 + 1]);
}

// c1[b] = sum_n G[n] * e[n]
__global__ void __launch_bounds__(256) dot_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                  double* __restrict__ out, int L, int chunk) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const int i0 = blockIdx.x * chunk;
  int i1 = i0 + chunk;
  if (i1 > L) i1 = L;
  double s = 0;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) s += (double)x[(long)b * L + i] * y[(long)b * L + i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + b, red[0] + red[1] + red[2] + red[3]);
}

// de = gl[b] * ( a*G + c1 * da/de ),  da/de[n] = t[n]/(sig_t D) - (a/D) (2 e[n] + 2 eps (e[n]-mean_e)/(L-1))
__global__ void __launch_bounds__(256) mrl1_bwd_kernel(const float* __restrict__ t, const float* __restrict__ e,
                                                       const float* __restrict__ G, const double* __restrict__ sums,
                                                       const double* __restrict__ c1, const float* __restrict__ gl,
                                                       float* __restrict__ de, int L, double eps,
                                                       const float* __restrict__ loss, int B) {
  const int b = blockIdx.y;
  if (loss) {            // NaN-loss guard of d_model.py:75-77: the step runs on zero gradients
    bool bad = false;
    for (int j = 0; j < B; ++j) bad |= isnan(loss[j]);
    if (bad) {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) de[(long)b * L + i] = 0.f;
      return;
    }
  }
  const Mrl1Scalars sc = mrl1_scalars(sums + (long)b * 5, L, eps);
  const double a = sc.set / (sc.sig_t * sc.D);
  const float k_t = (float)(c1[b] / (sc.sig_t * sc.D));
  const float k_e = (float)(c1[b] * a / sc.D);
  const float me = (float)sc.mean_e, ke2 = (float)(2.0 * eps / (L - 1));
  const float up = gl[b];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
    const long o = (long)b * L + i;
    const float ev = e[o];
    de[o] = up * (sc.a * G[o] + k_t * t[o] - k_e * (2.f * ev + ke2 * (ev - me)));
  }
}

// SI-SNR loss = 10 log10((1-coh)/coh), zero-mean, unit-norm (norm clamped at 1e-6)
__global__ void sisnr_final_kernel(const double* __restrict__ sums, float* __restrict__ loss, int B, int L) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* s = sums + (long)b * 5;
  const double srr = s[1] - s[0] * s[0] / L, sii = s[3] - s[2] * s[2] / L, sri = s[4] - s[0] * s[2] / L;
  const double nr = fmax(sqrt(fmax(srr, 0.0)), 1e-6), ni = fmax(sqrt(fmax(sii, 0.0)), 1e-6);
  const double c = sri / (nr * ni);
  const double coh = c * c;
  loss[b] = (float)(10.0 * log10((1.0 - coh) / coh));
}

// ---- optimizer ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ g, double* __restrict__ out, long n) {
  __shared__ double red[4];
  double s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = g[i];
    s += v * v;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.AdamW.step(); grads zeroed on the way out.
// A non-finite gradient norm skips the update (the reference's NaN guard, d_model.py:48-57).
__global__ void __launch_bounds__(256) clip_adamw_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const double* __restrict__ normsq, float max_norm, float lr,
                                                         float beta1, float beta2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float grad_scale, int zero_grad) {
  const double nsq = *normsq * (double)grad_scale * (double)grad_scale;
  const bool finite = isfinite(nsq);
  float coef = grad_scale;
  if (max_norm > 0.f) {
    const float c = max_norm / ((float)sqrt(nsq) + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (finite) {
      const float gr = g[i] * coef;
      float pv = p[i] * (1.f - lr * wd);
      const float mv = m[i] + (gr - m[i]) * (1.f - beta1);
      const float vv = v[i] * beta2 + gr * gr * (1.f - beta2);
      pv -= step * mv / (sqrtf(vv) / bc2_sqrt + eps);
      p[i] = pv; m[i] = mv; v[i] = vv;
    }
    if (zero_grad) g[i] = 0.f;
  }
}

// per-slot bookkeeping of the slotted AdamW step: a slot whose `used` flag is set (slot 0 always) advances its step
// count and gets its bias corrections; the others get {0, 0} = "leave these elements alone"
__global__ void adamw_slots_kernel(const double* __restrict__ normsq, float grad_scale, const float* __restrict__ used,
                                   int* __restrict__ steps, float* __restrict__ bias_corr, int n_slot,
                                   const unsigned* __restrict__ skip_flag, float beta1, float beta2) {
  const int s = threadIdx.x;
  if (s >= n_slot) return;
  const double nsq = *normsq * (double)grad_scale * (double)grad_scale;
  const bool go = isfinite(nsq) && !(skip_flag && *skip_flag != 0u);
  float b1 = 0.f, b2 = 0.f;
  if (go && (s == 0 || used[s] > 0.f)) {
    const int st = steps[s] + 1;
    steps[s] = st;
    b1 = 1.f - powf(beta1, (float)st);
    b2 = sqrtf(1.f - powf(beta2, (float)st));
  }
  bias_corr[2 * s] = b1;
  bias_corr[2 * s + 1] = b2;
}

__global__ void __launch_bounds__(256) clip_adamw_slots_kernel(float* __restrict__ p, float* __restrict__ g,
                                                               float* __restrict__ m, float* __restrict__ v, long n,
                                                               const double* __restrict__ normsq, float max_norm, float lr,
                                                               float beta1, float beta2, float eps, float wd,
                                                               const unsigned char* __restrict__ slot,
                                                               const float* __restrict__ bias_corr, int n_slot,
                                                               float grad_scale, int zero_grad) {
  __shared__ float bc[2 * 256];
  for (int i = threadIdx.x; i < 2 * n_slot; i += blockDim.x) bc[i] = bias_corr[i];
  __syncthreads();
  const double nsq = *normsq * (double)grad_scale * (double)grad_scale;
  float coef = grad_scale;
  if (max_norm > 0.f) {
    const float c = max_norm / ((float)sqrt(nsq) + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int s = slot[i];
    const float bc1 = bc[2 * s], bc2s = bc[2 * s + 1];
    if (bc1 > 0.f) {
      const float gr = g[i] * coef;
      float pv = p[i] * (1.f - lr * wd);
      const float mv = m[i] + (gr - m[i]) * (1.f - beta1);
      const float vv = v[i] * beta2 + gr * gr * (1.f - beta2);
      pv -= (lr / bc1) * mv / (sqrtf(vv) / bc2s + eps);
      p[i] = pv; m[i] = mv; v[i] = vv;
    }
    if (zero_grad) g[i] = 0.f;
  }
}

__global__ void fill_used_kernel(float* __restrict__ x, int n, int n_used) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = i < n_used ? 1.f : 0.f;
}

__global__ void zero_f32_kernel(float* __restrict__ x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 0.f;
}

}  // namespace urse

using namespace urse;

static std::once_flag g_loss_lds_once;

extern "C" int urse_pair_sums(const float* target, const float* estimate, double* sums, int B, int L, void* stream) {
  URSE_CHECK_ARG(target && estimate && sums && B > 0 && L > 1, "urse_pair_sums: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sums, 0, sizeof(double) * 5 * B, st);
  const int chunk = 4096;
  hipLaunchKernelGGL(pair_sums_kernel, dim3(ceil_div(L, chunk), B), dim3(256), 0, st, target, estimate, sums, L, chunk);
  URSE_CHECK_LAUNCH("urse_pair_sums");
  return URSE_OK;
}

extern "C" int urse_mrl1_loss_fwd(const float* target, const float* estimate, float* loss, float* G, double* sums,
                                  double* acc, int B, int L, const int32_t* windows, int n_windows, float eps,
                                  float td_weight, void* stream) {
  URSE_CHECK_ARG(target && estimate && loss && sums && acc && windows && B > 0 && L > 1 && n_windows > 0 &&
                     n_windows <= 8,
                 "urse_mrl1_loss_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::call_once(g_loss_lds_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mrl1_spec_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mrl1_spec_kernel<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  });
  int rc = urse_pair_sums(target, estimate, sums, B, L, stream);
  if (rc) return rc;
  (void)hipMemsetAsync(acc, 0, sizeof(double) * 2 * B, st);
  const int chunk = 4096;
  hipLaunchKernelGGL(mrl1_td_kernel, dim3(ceil_div(L, chunk), B), dim3(256), 0, st, target, estimate, sums, acc, G, L,
                     chunk, (double)eps, td_weight);
  const float w_spec = (1.f - td_weight) / (float)n_windows;
  for (int i = 0; i < n_windows; ++i) {
    const int n = windows[i], hop = n / 2;
    URSE_CHECK_ARG(n >= 4 && n % 2 == 0 && n / 2 < L, "urse_mrl1_loss_fwd: bad window %d", n);
    LossTables tb;
    rc = loss_tables(n, &tb);
    if (rc) return rc;
    const int T = L / hop + 1;
    const bool no_reg = getenv("URSE_MRL1_NO_REG_FFT") != nullptr;          // (A/B switch: the LDS-pass kernel)
    if (!no_reg && (n == 256 || n == 512 || n == 768 || n == 1024)) {
      // 256 = 16 x 16 (a quarter wave per frame, 16 frames per workgroup), 512 = 32 x 16, 768 = 32 x 24, 1024 = 32 x 32 (half waves, 8 frames)
      const int r1 = n == 256 ? 16 : 32, nfr = 256 / r1;
      const size_t lds = (size_t)(n + nfr * (n + r1)) * 8;
      dim3 grid(ceil_div(T, nfr), B);
#define URSE_MRL1_REG(R1_, R2_)                                                                                                           \
  {                                                                                                                                       \
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(mrl1_spec_reg_kernel<R1_, R2_, true>),                    \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024),                                \
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mrl1_spec_reg_kernel<R1_, R2_, false>),                   \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), true);                         \
    (void)once;                                                                                                                           \
    if (G) hipLaunchKernelGGL((mrl1_spec_reg_kernel<R1_, R2_, true>), grid, dim3(256), lds, st, target, estimate, sums, acc, G, L, T,     \
                              tb.tw, (double)eps, w_spec);                                                                                \
    else hipLaunchKernelGGL((mrl1_spec_reg_kernel<R1_, R2_, false>), grid, dim3(256), lds, st, target, estimate, sums, acc, G, L, T,      \
                            tb.tw, (double)eps, w_spec);                                                                                  \
  }
      if (n == 256) URSE_MRL1_REG(16, 16) else if (n == 512) URSE_MRL1_REG(32, 16) else if (n == 768) URSE_MRL1_REG(32, 24) else URSE_MRL1_REG(32, 32)
#undef URSE_MRL1_REG
      continue;
    }
    int NF = 8;
    while (NF > 2 && (size_t)(2 * NF + 1) * n * 8 > 72 * 1024) NF -= 2;
    const size_t lds = (size_t)(2 * NF + 1) * n * 8;
    dim3 grid(ceil_div(T, NF), B);
    if (G)
      hipLaunchKernelGGL(mrl1_spec_kernel<true>, grid, dim3(256), lds, st, target, estimate, sums, acc, G, L, T,
                         tb.plan, hop, tb.tw, NF, (double)eps, w_spec);
    else
      hipLaunchKernelGGL(mrl1_spec_kernel<false>, grid, dim3(256), lds, st, target, estimate, sums, acc, G, L, T,
                         tb.plan, hop, tb.tw, NF, (double)eps, w_spec);
  }
  hipLaunchKernelGGL(mrl1_final_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, acc, loss, B, td_weight, w_spec);
  URSE_CHECK_LAUNCH("urse_mrl1_loss_fwd");
  return URSE_OK;
}

extern "C" int urse_mrl1_loss_bwd(const float* target, const float* estimate, const float* G, const double* sums,
                                  const float* grad_loss, float* grad_estimate, double* c1, int B, int L, float eps,
                                  void* stream) {
  URSE_CHECK_ARG(target && estimate && G && sums && grad_loss && grad_estimate && c1 && B > 0 && L > 1,
                 "urse_mrl1_loss_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(c1, 0, sizeof(double) * B, st);
  const int chunk = 4096;
  hipLaunchKernelGGL(dot_kernel, dim3(ceil_div(L, chunk), B), dim3(256), 0, st, G, estimate, c1, L, chunk);
  hipLaunchKernelGGL(mrl1_bwd_kernel, dim3(ceil_div(L, 1024), B), dim3(256), 0, st, target, estimate, G, sums, c1,
                     grad_loss, grad_estimate, L, (double)eps, (const float*)nullptr, B);
  URSE_CHECK_LAUNCH("urse_mrl1_loss_bwd");
  return URSE_OK;
}

extern "C" int urse_mrl1_loss_bwd_guarded(const float* target, const float* estimate, const float* G, const double* sums,
                                          const float* grad_loss, const float* loss, float* grad_estimate, double* c1,
                                          int B, int L, float eps, void* stream) {
  URSE_CHECK_ARG(target && estimate && G && sums && grad_loss && loss && grad_estimate && c1 && B > 0 && L > 1,
                 "urse_mrl1_loss_bwd_guarded: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(c1, 0, sizeof(double) * B, st);
  const int chunk = 4096;
  hipLaunchKernelGGL(dot_kernel, dim3(ceil_div(L, chunk), B), dim3(256), 0, st, G, estimate, c1, L, chunk);
  hipLaunchKernelGGL(mrl1_bwd_kernel, dim3(ceil_div(L, 1024), B), dim3(256), 0, st, target, estimate, G, sums, c1,
                     grad_loss, grad_estimate, L, (double)eps, loss, B);
  URSE_CHECK_LAUNCH("urse_mrl1_loss_bwd_guarded");
  return URSE_OK;
}

extern "C" int urse_sisnr_fwd(const float* ref, const float* inf, float* loss, double* sums, int B, int L,
                              void* stream) {
  int rc = urse_pair_sums(ref, inf, sums, B, L, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(sisnr_final_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, sums, loss, B, L);
  URSE_CHECK_LAUNCH("urse_sisnr_fwd");
  return URSE_OK;
}

extern "C" int urse_grad_sumsq(const float* g, double* out, int64_t n, void* stream) {
  URSE_CHECK_ARG(g && out && n > 0, "urse_grad_sumsq: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(out, 0, sizeof(double), st);
  hipLaunchKernelGGL(sumsq_kernel, dim3(1024), dim3(256), 0, st, g, out, (long)n);
  URSE_CHECK_LAUNCH("urse_grad_sumsq");
  return URSE_OK;
}

extern "C" int urse_clip_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                    const double* normsq, float max_norm, float lr, float beta1, float beta2,
                                    float eps, float weight_decay, int step, float grad_scale, int zero_grad,
                                    void* stream) {
  URSE_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && normsq && n > 0 && step >= 1,
                 "urse_clip_adamw_step: bad argument");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(clip_adamw_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, (long)n, normsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale,
                     zero_grad);
  URSE_CHECK_LAUNCH("urse_clip_adamw_step");
  return URSE_OK;
}

extern "C" int urse_clip_adamw_step_slots(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                          const double* normsq, float max_norm, float lr, float beta1, float beta2,
                                          float eps, float weight_decay, const uint8_t* slot, float* used, int32_t* steps,
                                          float* bias_corr, int n_slot, const uint32_t* skip_flag, float grad_scale,
                                          int zero_grad, void* stream) {
  URSE_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && normsq && slot && used && steps && bias_corr && n > 0 &&
                     n_slot >= 1 && n_slot <= 256,
                 "urse_clip_adamw_step_slots: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_slots_kernel, dim3(1), dim3(256), 0, st, normsq, grad_scale, used, steps, bias_corr, n_slot,
                     skip_flag, beta1, beta2);
  hipLaunchKernelGGL(clip_adamw_slots_kernel, dim3(2048), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, (long)n,
                     normsq, max_norm, lr, beta1, beta2, eps, weight_decay, slot, bias_corr, n_slot, grad_scale, zero_grad);
  if (zero_grad) hipLaunchKernelGGL(zero_f32_kernel, dim3(1), dim3(256), 0, st, used, n_slot);
  URSE_CHECK_LAUNCH("urse_clip_adamw_step_slots");
  return URSE_OK;
}

extern "C" int urse_fill_used_flags(float* used, int n_slot, int n_used, void* stream) {
  URSE_CHECK_ARG(used && n_slot >= 1 && n_slot <= 256 && n_used >= 0, "urse_fill_used_flags: bad argument");
  hipLaunchKernelGGL(fill_used_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, used, n_slot, n_used);
  URSE_CHECK_LAUNCH("urse_fill_used_flags");
  return URSE_OK;
}
