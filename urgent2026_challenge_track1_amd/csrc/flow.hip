// BSRNN-Flow specific kernels: spectral exponent transform, flow-matching state preparation, GradDecoder tail
// (Conv2d(16->4, 5x5, pad 2) + GLU over the (F,T) plane, complex m*x_t + r), flow-matching loss, EMA.
// Reference: baseline_code/models/bsrnn_flowse.py:103-168 (GradDecoder), :311-315 (output), flow_model.py:122-132
// (_loss), :159-172 (xt / conditional vector field), espnet STFTEncoder 'exponent' transform,
// torch_ema update.  Feature maps are channel-last [B, T, F, C]; all HBM-bound / small.
#include "urse_common.h"

namespace urse {

static int fgrid(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  return g < 1 ? 1 : (int)g;
}

// |X|^e * e^{j angle X} * factor   (inverse: X/factor, then |.|^(1/e))
__global__ void __launch_bounds__(256) spec_transform_kernel(const float2* __restrict__ x, float2* __restrict__ y,
                                                             long n, float exponent, float factor, int inverse) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float2 v = x[i];
    if (inverse) { v.x /= factor; v.y /= factor; }
    const float mag = sqrtf(v.x * v.x + v.y * v.y);
    const float e = inverse ? 1.0f / exponent : exponent;
    float s = 0.f;
    if (mag > 0.f) s = (e == 1.0f) ? 1.0f : powf(mag, e - 1.0f);
    if (!inverse) s *= factor;
    y[i] = make_float2(v.x * s, v.y * s);
  }
}

// xt = (1-t) x0 + t y + sigma(t) z ;  cvf = (smax - smin) z + (y - x0)       (complex, per_b elements per utterance)
__global__ void __launch_bounds__(256) flow_prepare_kernel(const float2* __restrict__ x0, const float2* __restrict__ y,
                                                           const float2* __restrict__ z, const float* __restrict__ t,
                                                           float2* __restrict__ xt, float2* __restrict__ cvf,
                                                           long per_b, long n, float smin, float smax) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float tb = t[i / per_b];
    const float sg = (1.f - tb) * smin + tb * smax, ds = smax - smin;
    const float2 a = x0[i], b = y[i], c = z[i];
    xt[i] = make_float2((1.f - tb) * a.x + tb * b.x + sg * c.x, (1.f - tb) * a.y + tb * b.y + sg * c.y);
    if (cvf) cvf[i] = make_float2(ds * c.x + (b.x - a.x), ds * c.y + (b.y - a.y));
  }
}

// y = y + a * x (f32; the Euler update on the interleaved complex state)
__global__ void __launch_bounds__(256) axpy_kernel(const float* __restrict__ x, float* __restrict__ y, float a, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}

// Gaussian Fourier time embedding: out[b, :] = [sin(2 pi t_b W), cos(2 pi t_b W)]  (bsrnn_flowse.py:90-99)
__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ W, float* __restrict__ out, int B,
                            int half) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  const int b = idx / half, j = idx - b * half;
  const float p = t[b] * W[j] * 2.0f * 3.14159265358979323846f;
  out[b * 2 * half + j] = sinf(p);
  out[b * 2 * half + half + j] = cosf(p);
}

// ---- Conv2d(16 -> 4, 5x5, pad 2) over (F, T) on channel-last maps --------------------------------------------
constexpr int CT = 8, CF = 32, CIN = 16, COUT = 4, KS = 5, HALO = 2;
constexpr int TT = CT + 2 * HALO, TF = CF + 2 * HALO;

// U f32 [B,T,F,16] -> pre f32 [B,T,F,4]; W [4][16][5(f)][5(t)], bias [4]
__global__ void __launch_bounds__(256) conv5x5_fwd_kernel(const float* __restrict__ U, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ pre,
                                                          int T, int F) {
  __shared__ float tile[CIN][TT][TF + 1];
  __shared__ float w[COUT * CIN * KS * KS];
  const int b = blockIdx.z, t0 = blockIdx.y * CT, f0 = blockIdx.x * CF, tid = threadIdx.x;
  for (int i = tid; i < COUT * CIN * KS * KS; i += 256) w[i] = W[i];
  for (int idx = tid; idx < TT * TF * CIN; idx += 256) {
    const int ic = idx & 15, pos = idx >> 4;
    const int ff = pos % TF, tt = pos / TF;
    const int t = t0 + tt - HALO, f = f0 + ff - HALO;
    float v = 0.f;
    if (t >= 0 && t < T && f >= 0 && f < F) v = U[(((long)b * T + t) * F + f) * CIN + ic];
    tile[ic][tt][ff] = v;
  }
  __syncthreads();
  const int ff = tid & 31, tt = tid >> 5;
  const int t = t0 + tt, f = f0 + ff;
  float acc[COUT] = {bias[0], bias[1], bias[2], bias[3]};
  for (int ic = 0; ic < CIN; ++ic)
#pragma unroll
    for (int df = 0; df < KS; ++df)
#pragma unroll
      for (int dt = 0; dt < KS; ++dt) {
        const float u = tile[ic][tt + dt][ff + df];
#pragma unroll
        for (int oc = 0; oc < COUT; ++oc) acc[oc] += w[((oc * CIN + ic) * KS + df) * KS + dt] * u;
      }
  if (t < T && f < F)
    *reinterpret_cast<float4*>(pre + (((long)b * T + t) * F + f) * COUT) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// dU[b,t,f,ic] = sum_{oc,df,dt} W[oc][ic][df][dt] * dpre[b, t-dt+2, f-df+2, oc]
__global__ void __launch_bounds__(256) conv5x5_bwd_data_kernel(const float* __restrict__ dpre, const float* __restrict__ W,
                                                               float* __restrict__ dU, int T, int F) {
  __shared__ float tile[COUT][TT][TF + 1];
  __shared__ float w[COUT * CIN * KS * KS];
  const int b = blockIdx.z, t0 = blockIdx.y * CT, f0 = blockIdx.x * CF, tid = threadIdx.x;
  for (int i = tid; i < COUT * CIN * KS * KS; i += 256) w[i] = W[i];
  for (int idx = tid; idx < TT * TF * COUT; idx += 256) {
    const int oc = idx & 3, pos = idx >> 2;
    const int ff = pos % TF, tt = pos / TF;
    const int t = t0 + tt - HALO, f = f0 + ff - HALO;
    float v = 0.f;
    if (t >= 0 && t < T && f >= 0 && f < F) v = dpre[(((long)b * T + t) * F + f) * COUT + oc];
    tile[oc][tt][ff] = v;
  }
  __syncthreads();
  const int ff = tid & 31, tt = tid >> 5;
  const int t = t0 + tt, f = f0 + ff;
  float acc[CIN];
#pragma unroll
  for (int ic = 0; ic < CIN; ++ic) acc[ic] = 0.f;
  for (int oc = 0; oc < COUT; ++oc)
#pragma unroll
    for (int df = 0; df < KS; ++df)
#pragma unroll
      for (int dt = 0; dt < KS; ++dt) {
        const float g = tile[oc][tt + 2 * HALO - dt][ff + 2 * HALO - df];
#pragma unroll
        for (int ic = 0; ic < CIN; ++ic) acc[ic] += w[((oc * CIN + ic) * KS + df) * KS + dt] * g;
      }
  if (t < T && f < F) {
    float* o = dU + (((long)b * T + t) * F + f) * CIN;
#pragma unroll
    for (int ic = 0; ic < CIN; ic += 4)
      *reinterpret_cast<float4*>(o + ic) = make_float4(acc[ic], acc[ic + 1], acc[ic + 2], acc[ic + 3]);
  }
}

// dW[oc][ic][df][dt] += sum_{b,t,f} dpre[b,t,f,oc] * U[b,t+dt-2,f+df-2,ic] ; dbias[oc] += sum dpre
__global__ void __launch_bounds__(256) conv5x5_bwd_weight_kernel(const float* __restrict__ U, const float* __restrict__ dpre,
                                                                 float* __restrict__ dW, float* __restrict__ dbias,
                                                                 int T, int F) {
  __shared__ float tile[CIN][TT][TF + 1];
  __shared__ float g[COUT][CT][CF];
  const int b = blockIdx.z, t0 = blockIdx.y * CT, f0 = blockIdx.x * CF, tid = threadIdx.x;
  for (int idx = tid; idx < TT * TF * CIN; idx += 256) {
    const int ic = idx & 15, pos = idx >> 4;
    const int ff = pos % TF, tt = pos / TF;
    const int t = t0 + tt - HALO, f = f0 + ff - HALO;
    float v = 0.f;
    if (t >= 0 && t < T && f >= 0 && f < F) v = U[(((long)b * T + t) * F + f) * CIN + ic];
    tile[ic][tt][ff] = v;
  }
  for (int idx = tid; idx < CT * CF * COUT; idx += 256) {
    const int oc = idx & 3, pos = idx >> 2;
    const int ff = pos % CF, tt = pos / CF;
    const int t = t0 + tt, f = f0 + ff;
    g[oc][tt][ff] = (t < T && f < F) ? dpre[(((long)b * T + t) * F + f) * COUT + oc] : 0.f;
  }
  __syncthreads();
  // 1600 weights over 256 threads: thread handles weights tid, tid+256, ...
  for (int wi = tid; wi < COUT * CIN * KS * KS; wi += 256) {
    const int dt = wi % KS, df = (wi / KS) % KS, ic = (wi / (KS * KS)) % CIN, oc = wi / (KS * KS * CIN);
    float s = 0.f;
    for (int tt = 0; tt < CT; ++tt)
      for (int ff = 0; ff < CF; ++ff) s += g[oc][tt][ff] * tile[ic][tt + dt][ff + df];
    atomicAdd(dW + wi, s);
  }
  if (tid < COUT) {
    float s = 0.f;
    for (int tt = 0; tt < CT; ++tt)
      for (int ff = 0; ff < CF; ++ff) s += g[tid][tt][ff];
    atomicAdd(dbias + tid, s);
  }
}

// out = GLU4(pre_m) * x + GLU4(pre_r):  pre [.., 4] -> (c0*sig(c2), c1*sig(c3)) = (re, im);  sign: out = sgn * (m x + r)
__global__ void __launch_bounds__(256) glu4_apply_kernel(const float4* __restrict__ pm, const float4* __restrict__ pr,
                                                         const float2* __restrict__ x, float2* __restrict__ out, long rows,
                                                         int F, int Fs, float sgn) {
  const long n = rows * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / F;
    const long j = row * Fs + (i - row * F);   // pre maps are Fs >= F wide (zero-padded last band), x / out are F wide
    const float4 a = pm[j], c = pr[j];
    const float2 m = make_float2(a.x * sigmoidf_(a.z), a.y * sigmoidf_(a.w));
    const float2 r = make_float2(c.x * sigmoidf_(c.z), c.y * sigmoidf_(c.w));
    const float2 xv = x[i];
    out[i] = make_float2(sgn * (m.x * xv.x - m.y * xv.y + r.x), sgn * (m.x * xv.y + m.y * xv.x + r.y));
  }
}

__global__ void __launch_bounds__(256) glu4_apply_bwd_kernel(const float4* __restrict__ pm, const float4* __restrict__ pr,
                                                             const float2* __restrict__ x, const float2* __restrict__ dout,
                                                             float4* __restrict__ dpm, float4* __restrict__ dpr, long rows,
                                                             int F, int Fs, float sgn) {
  const long n = rows * Fs;
  for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
    const long row = j / Fs;
    const int f = (int)(j - row * Fs);
    if (f >= F) { dpm[j] = make_float4(0.f, 0.f, 0.f, 0.f); dpr[j] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    const long i = row * F + f;
    const float4 a = pm[j], c = pr[j];
    const float2 xv = x[i];
    const float2 g = make_float2(sgn * dout[i].x, sgn * dout[i].y);
    const float2 dm = make_float2(g.x * xv.x + g.y * xv.y, g.y * xv.x - g.x * xv.y);   // g * conj(x)
    const float s2 = sigmoidf_(a.z), s3 = sigmoidf_(a.w), q2 = sigmoidf_(c.z), q3 = sigmoidf_(c.w);
    dpm[j] = make_float4(dm.x * s2, dm.y * s3, dm.x * a.x * s2 * (1.f - s2), dm.y * a.y * s3 * (1.f - s3));
    dpr[j] = make_float4(g.x * q2, g.y * q3, g.x * c.x * q2 * (1.f - q2), g.y * c.y * q3 * (1.f - q3));
  }
}

// dst (TO, pitch ldd) = dU * (1 - U^2): tanh backward of the decoder feature map, cast to the GEMM operand type
template <typename TO>
__global__ void __launch_bounds__(256) tanh_bwd_pack_kernel(const float* __restrict__ dU, const float* __restrict__ U,
                                                            TO* __restrict__ dst, long rows, int n, long ldd) {
  const long total = rows * n;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / n;
    const int c = (int)(idx - r * n);
    const float u = U[idx];
    dst[r * ldd + c] = from_f32<TO>(dU[idx] * (1.f - u * u));
  }
}

// loss_b = 0.5 * sum |vf - cvf|^2 ; grad = (vf - cvf) * scale
__global__ void __launch_bounds__(256) flow_loss_kernel(const float2* __restrict__ vf, const float2* __restrict__ cvf,
                                                        double* __restrict__ loss, float2* __restrict__ grad,
                                                        long per_b, float scale) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  double s = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_b; i += (long)gridDim.x * blockDim.x) {
    const long o = (long)b * per_b + i;
    const float dx = vf[o].x - cvf[o].x, dy = vf[o].y - cvf[o].y;
    s += (double)dx * dx + (double)dy * dy;
    if (grad) grad[o] = make_float2(dx * scale, dy * scale);
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + b, 0.5 * (red[0] + red[1] + red[2] + red[3]));
}

// torch_ema: shadow -= (1 - d) * (shadow - p)
__global__ void __launch_bounds__(256) ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, float omd,
                                                  long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    shadow[i] -= omd * (shadow[i] - p[i]);
}

}  // namespace urse

using namespace urse;

extern "C" int urse_spec_transform(const float* x, float* y, int64_t n_complex, float exponent, float factor, int inverse,
                                   void* stream) {
  URSE_CHECK_ARG(x && y && n_complex > 0 && exponent > 0.f && factor > 0.f, "urse_spec_transform: bad argument");
  hipLaunchKernelGGL(spec_transform_kernel, dim3(fgrid(n_complex)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, (float2*)y, (long)n_complex, exponent, factor, inverse);
  URSE_CHECK_LAUNCH("urse_spec_transform");
  return URSE_OK;
}

extern "C" int urse_flow_prepare(const float* x0, const float* y, const float* z, const float* t, float* xt, float* cvf,
                                 int B, int64_t per_b, float sigma_min, float sigma_max, void* stream) {
  URSE_CHECK_ARG(x0 && y && z && t && xt && B > 0 && per_b > 0, "urse_flow_prepare: bad argument");
  const long n = (long)B * per_b;
  hipLaunchKernelGGL(flow_prepare_kernel, dim3(fgrid(n)), dim3(256), 0, (hipStream_t)stream, (const float2*)x0,
                     (const float2*)y, (const float2*)z, t, (float2*)xt, (float2*)cvf, (long)per_b, n, sigma_min,
                     sigma_max);
  URSE_CHECK_LAUNCH("urse_flow_prepare");
  return URSE_OK;
}

extern "C" int urse_axpy(const float* x, float* y, float a, int64_t n, void* stream) {
  URSE_CHECK_ARG(x && y && n > 0, "urse_axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(fgrid(n)), dim3(256), 0, (hipStream_t)stream, x, y, a, (long)n);
  URSE_CHECK_LAUNCH("urse_axpy");
  return URSE_OK;
}

extern "C" int urse_time_embedding(const float* t, const float* W, float* out, int B, int half, void* stream) {
  URSE_CHECK_ARG(t && W && out && B > 0 && half > 0, "urse_time_embedding: bad argument");
  hipLaunchKernelGGL(temb_kernel, dim3(ceil_div((long)B * half, 256)), dim3(256), 0, (hipStream_t)stream, t, W, out, B,
                     half);
  URSE_CHECK_LAUNCH("urse_time_embedding");
  return URSE_OK;
}

extern "C" int urse_conv5x5_fwd(const float* U, const float* W, const float* bias, float* pre, int B, int T, int F,
                                void* stream) {
  URSE_CHECK_ARG(U && W && bias && pre && B > 0 && T > 0 && F > 0, "urse_conv5x5_fwd: bad argument");
  hipLaunchKernelGGL(conv5x5_fwd_kernel, dim3(ceil_div(F, CF), ceil_div(T, CT), B), dim3(256), 0, (hipStream_t)stream, U,
                     W, bias, pre, T, F);
  URSE_CHECK_LAUNCH("urse_conv5x5_fwd");
  return URSE_OK;
}

extern "C" int urse_conv5x5_bwd(const float* U, const float* W, const float* dpre, float* dU, float* dW, float* dbias,
                                int B, int T, int F, void* stream) {
  URSE_CHECK_ARG(U && W && dpre && dU && dW && dbias && B > 0 && T > 0 && F > 0, "urse_conv5x5_bwd: bad argument");
  dim3 grid(ceil_div(F, CF), ceil_div(T, CT), B);
  hipLaunchKernelGGL(conv5x5_bwd_data_kernel, grid, dim3(256), 0, (hipStream_t)stream, dpre, W, dU, T, F);
  hipLaunchKernelGGL(conv5x5_bwd_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, U, dpre, dW, dbias, T, F);
  URSE_CHECK_LAUNCH("urse_conv5x5_bwd");
  return URSE_OK;
}

extern "C" int urse_glu4_apply_fwd(const float* pre_m, const float* pre_r, const float* x, float* out, int64_t rows,
                                   int F, int Fs, float sign, void* stream) {
  URSE_CHECK_ARG(pre_m && pre_r && x && out && rows > 0 && F > 0 && Fs >= F, "urse_glu4_apply_fwd: bad argument");
  hipLaunchKernelGGL(glu4_apply_kernel, dim3(fgrid(rows * F)), dim3(256), 0, (hipStream_t)stream, (const float4*)pre_m,
                     (const float4*)pre_r, (const float2*)x, (float2*)out, (long)rows, F, Fs, sign);
  URSE_CHECK_LAUNCH("urse_glu4_apply_fwd");
  return URSE_OK;
}

extern "C" int urse_glu4_apply_bwd(const float* pre_m, const float* pre_r, const float* x, const float* dout,
                                   float* dpre_m, float* dpre_r, int64_t rows, int F, int Fs, float sign,
                                   void* stream) {
  URSE_CHECK_ARG(pre_m && pre_r && x && dout && dpre_m && dpre_r && rows > 0 && F > 0 && Fs >= F,
                 "urse_glu4_apply_bwd: bad argument");
  hipLaunchKernelGGL(glu4_apply_bwd_kernel, dim3(fgrid(rows * Fs)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)pre_m, (const float4*)pre_r, (const float2*)x, (const float2*)dout, (float4*)dpre_m,
                     (float4*)dpre_r, (long)rows, F, Fs, sign);
  URSE_CHECK_LAUNCH("urse_glu4_apply_bwd");
  return URSE_OK;
}

extern "C" int urse_tanh_bwd_pack(const float* dU, const float* U, void* dst, int64_t rows, int n, int64_t ldd,
                                  int out_dtype, void* stream) {
  URSE_CHECK_ARG(dU && U && dst && rows > 0 && n > 0 && ldd >= n, "urse_tanh_bwd_pack: bad argument");
  dim3 g(fgrid(rows * n)), b(256);
  if (out_dtype == URSE_BF16)
    hipLaunchKernelGGL(tanh_bwd_pack_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, dU, U, (bf16_t*)dst, (long)rows, n,
                       (long)ldd);
  else
    hipLaunchKernelGGL(tanh_bwd_pack_kernel<float>, g, b, 0, (hipStream_t)stream, dU, U, (float*)dst, (long)rows, n,
                       (long)ldd);
  URSE_CHECK_LAUNCH("urse_tanh_bwd_pack");
  return URSE_OK;
}

extern "C" int urse_flow_loss(const float* vf, const float* cvf, double* loss, float* grad, int B, int64_t per_b,
                              float grad_scale, void* stream) {
  URSE_CHECK_ARG(vf && cvf && loss && B > 0 && per_b > 0, "urse_flow_loss: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(loss, 0, sizeof(double) * B, st);
  hipLaunchKernelGGL(flow_loss_kernel, dim3(ceil_div(per_b, 256 * 8), B), dim3(256), 0, st, (const float2*)vf,
                     (const float2*)cvf, loss, (float2*)grad, (long)per_b, grad_scale);
  URSE_CHECK_LAUNCH("urse_flow_loss");
  return URSE_OK;
}

extern "C" int urse_ema_update(float* shadow, const float* params, float one_minus_decay, int64_t n, void* stream) {
  URSE_CHECK_ARG(shadow && params && n > 0, "urse_ema_update: bad argument");
  hipLaunchKernelGGL(ema_kernel, dim3(fgrid(n)), dim3(256), 0, (hipStream_t)stream, shadow, params, one_minus_decay,
                     (long)n);
  URSE_CHECK_LAUNCH("urse_ema_update");
  return URSE_OK;
}
