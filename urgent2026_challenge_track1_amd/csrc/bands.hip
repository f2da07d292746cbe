// Band-split front end and mask-decoder back end of BSRNN (the pieces around the grouped GEMMs).
//
//  * bandsplit_norm: per band GroupNorm(1, 2*sb) of the (re,im)-interleaved sub-band slice of the
//    spectrum (zero-padded when the band sticks out of F, exactly as F.pad before the norm), written as
//    the zero-padded GEMM operand xnb[B*T, sum_k kpad(2*sb_k)].
//  * glu_mask_apply: GLU(dim=channel) of the two decoder heads + complex m*x + r, fused; and its adjoint.
// espnet2 BandSplit / MaskDecoder; in-tree twin baseline_code/models/bsrnn_flowse.py:63-86
// (band loop, padding, norm), :311-315 (complex mask apply).  All HBM-bound elementwise / reduction work.
#include "urse_common.h"

namespace urse {

struct Band {  // one row of the int32 [K, 8] band table built by the host
  int f0, sb, xoff, xpad;   // first bin, bins, column offset / padded width in xnb
  int goff, poff, ppad, r0; // offset into concatenated gamma/beta (2*sb each), column offset / padded width in `pre`
};

__global__ void __launch_bounds__(256) bandsplit_stats_kernel(const float* __restrict__ spec, const Band* __restrict__ bands,
                                                              double* __restrict__ stats, int T, int F) {
  __shared__ double red[8];
  const int k = blockIdx.x, b = blockIdx.y, K = gridDim.x;
  const Band bd = bands[k];
  int valid = F - bd.f0;
  if (valid > bd.sb) valid = bd.sb;
  const int w = 2 * valid;
  const float* base = spec + ((long)b * T) * 2 * F + 2 * bd.f0;
  double s = 0.0, q = 0.0;
  for (int idx = threadIdx.x; idx < T * w; idx += blockDim.x) {
    const int t = idx / w, c = idx - t * w;
    const float v = base[(long)t * 2 * F + c];
    s += v;
    q += (double)v * v;
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv] = s; red[4 + wv] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[((long)b * K + k) * 2] = red[0] + red[1] + red[2] + red[3];
    stats[((long)b * K + k) * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

template <typename TO>
__global__ void __launch_bounds__(256) bandsplit_apply_kernel(const float* __restrict__ spec, const Band* __restrict__ bands,
                                                              const double* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              TO* __restrict__ xnb, int B, int T, int F, int K, int ldx,
                                                              float eps, bf16_t* __restrict__ xnb2 = nullptr) {
  // blockIdx.x = row (b,t); threads sweep the padded columns band by band.  xnb2 (f16 forward mode, training): the rows once more in bf16
  // for the weight-gradient GEMM
  const long row = blockIdx.x;
  const int b = (int)(row / T);
  const float* src = spec + row * 2 * F;
  TO* dst = xnb + row * ldx;
  bf16_t* dst2 = xnb2 ? xnb2 + row * ldx : nullptr;
  for (int k = 0; k < K; ++k) {
    const Band bd = bands[k];
    const double cnt = (double)T * 2 * bd.sb;
    const double m = stats[((long)b * K + k) * 2] / cnt;
    double var = stats[((long)b * K + k) * 2 + 1] / cnt - m * m;
    if (var < 0) var = 0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int lim = 2 * (F - bd.f0);
    for (int c = threadIdx.x; c < bd.xpad; c += blockDim.x) {
      float v = 0.f;
      if (c < 2 * bd.sb) {
        const float x = (c < lim) ? src[2 * bd.f0 + c] : 0.f;
        v = (x - mean) * rstd * gamma[bd.goff + c] + beta[bd.goff + c];
      }
      dst[bd.xoff + c] = from_f32<TO>(v);
      if (dst2) dst2[bd.xoff + c] = f32_to_bf16(v);
    }
  }
}

// dgamma[c] += sum_{b,t} dxnb * xhat ; dbeta[c] += sum dxnb        grid (K, B, tchunks)
__global__ void __launch_bounds__(128) bandsplit_bwd_affine_kernel(const float* __restrict__ spec, const float* __restrict__ dxnb,
                                                                   const Band* __restrict__ bands,
                                                                   const double* __restrict__ stats,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   int T, int F, int ldx, float eps, int tchunk) {
  const int k = blockIdx.x, b = blockIdx.y, K = gridDim.x;
  const Band bd = bands[k];
  const int c = threadIdx.x;
  if (c >= 2 * bd.sb) return;
  const double cnt = (double)T * 2 * bd.sb;
  const double m = stats[((long)b * K + k) * 2] / cnt;
  double var = stats[((long)b * K + k) * 2 + 1] / cnt - m * m;
  if (var < 0) var = 0;
  const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const bool in = c < 2 * (F - bd.f0);
  int t0 = blockIdx.z * tchunk, t1 = t0 + tchunk;
  if (t1 > T) t1 = T;
  float dg = 0.f, db = 0.f;
  for (int t = t0; t < t1; ++t) {
    const long row = (long)b * T + t;
    const float x = in ? spec[row * 2 * F + 2 * bd.f0 + c] : 0.f;
    const float d = dxnb[row * ldx + bd.xoff + c];
    dg += d * (x - mean) * rstd;
    db += d;
  }
  atomicAdd(dgamma + bd.goff + c, dg);
  atomicAdd(dbeta + bd.goff + c, db);
}

// out[b,t,f] = GLU(pre_m)[f] * x[b,t,f] + GLU(pre_r)[f]   (complex); f2k maps bin -> band (-1: beyond the used bands)
__global__ void __launch_bounds__(256) glu_mask_apply_kernel(const float* __restrict__ pre_m, const float* __restrict__ pre_r,
                                                             const float2* __restrict__ x, float2* __restrict__ out,
                                                             const Band* __restrict__ bands, const int* __restrict__ f2k,
                                                             long rows, int F, int ldp) {
  const long total = rows * F;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / F;
    const int f = (int)(idx - row * F);
    const int k = f2k[f];
    float2 o = make_float2(0.f, 0.f);
    if (k >= 0) {
      const Band bd = bands[k];
      const int c = 2 * (f - bd.f0);
      const float* pm = pre_m + row * ldp + bd.poff;
      const float* pr = pre_r + row * ldp + bd.poff;
      const float2 am = *reinterpret_cast<const float2*>(pm + c);
      const float2 gm = *reinterpret_cast<const float2*>(pm + 2 * bd.sb + c);
      const float2 ar = *reinterpret_cast<const float2*>(pr + c);
      const float2 gr = *reinterpret_cast<const float2*>(pr + 2 * bd.sb + c);
      const float2 m = make_float2(am.x * sigmoidf_(gm.x), am.y * sigmoidf_(gm.y));
      const float2 r = make_float2(ar.x * sigmoidf_(gr.x), ar.y * sigmoidf_(gr.y));
      const float2 xv = x[idx];
      o = make_float2(m.x * xv.x - m.y * xv.y + r.x, m.x * xv.y + m.y * xv.x + r.y);
    }
    out[idx] = o;
  }
}

// adjoint: dm = dout * conj(x), dr = dout, then GLU backward into the (TO, zero-padded) dpre operands
template <typename TO>
__global__ void __launch_bounds__(256) glu_mask_apply_bwd_kernel(const float* __restrict__ pre_m, const float* __restrict__ pre_r,
                                                                 const float2* __restrict__ x, const float2* __restrict__ dout,
                                                                 TO* __restrict__ dpre_m, TO* __restrict__ dpre_r,
                                                                 const Band* __restrict__ bands, const int* __restrict__ f2k,
                                                                 long rows, int F, int ldp) {
  const long total = rows * F;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / F;
    const int f = (int)(idx - row * F);
    const int k = f2k[f];
    if (k < 0) continue;
    const Band bd = bands[k];
    const int c = 2 * (f - bd.f0);
    const long o = row * ldp + bd.poff;
    const float2 xv = x[idx], g = dout[idx];
    // dm = g * conj(x)
    const float dm[2] = {g.x * xv.x + g.y * xv.y, g.y * xv.x - g.x * xv.y};
    const float dr[2] = {g.x, g.y};
#pragma unroll
    for (int ri = 0; ri < 2; ++ri) {
      const float a = pre_m[o + c + ri], s = sigmoidf_(pre_m[o + 2 * bd.sb + c + ri]);
      dpre_m[o + c + ri] = from_f32<TO>(dm[ri] * s);
      dpre_m[o + 2 * bd.sb + c + ri] = from_f32<TO>(dm[ri] * a * s * (1.f - s));
      const float a2 = pre_r[o + c + ri], s2 = sigmoidf_(pre_r[o + 2 * bd.sb + c + ri]);
      dpre_r[o + c + ri] = from_f32<TO>(dr[ri] * s2);
      dpre_r[o + 2 * bd.sb + c + ri] = from_f32<TO>(dr[ri] * a2 * s2 * (1.f - s2));
    }
  }
}

// y = a*x + b*y (f32), tiny helper for bias sums / gradient copies
__global__ void __launch_bounds__(256) axpby_kernel(const float* __restrict__ x, float* __restrict__ y, float a, float b,
                                                    long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

}  // namespace urse

using namespace urse;

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  return g < 1 ? 1 : (int)g;
}

extern "C" int urse_bandsplit_norm_fwd(const float* spec, const int32_t* bands, const float* gamma, const float* beta,
                                       void* xnb, double* stats, int B, int T, int F, int K, int ldx, float eps,
                                       int out_dtype, void* xnb_bf16, void* stream) {
  URSE_CHECK_ARG(spec && bands && gamma && beta && xnb && stats && B > 0 && T > 0 && F > 0 && K > 0,
                 "urse_bandsplit_norm_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bandsplit_stats_kernel, dim3(K, B), dim3(256), 0, st, spec, (const Band*)bands, stats, T, F);
  URSE_CHECK_ARG(!xnb_bf16 || out_dtype == URSE_F16, "urse_bandsplit_norm_fwd: the bf16 copy goes with f16 output only");
  if (out_dtype == URSE_BF16)
    hipLaunchKernelGGL(bandsplit_apply_kernel<bf16_t>, dim3(B * T), dim3(256), 0, st, spec, (const Band*)bands, stats,
                       gamma, beta, (bf16_t*)xnb, B, T, F, K, ldx, eps, (bf16_t*)nullptr);
  else if (out_dtype == URSE_F16)
    hipLaunchKernelGGL(bandsplit_apply_kernel<f16_t>, dim3(B * T), dim3(256), 0, st, spec, (const Band*)bands, stats,
                       gamma, beta, (f16_t*)xnb, B, T, F, K, ldx, eps, (bf16_t*)xnb_bf16);
  else
    hipLaunchKernelGGL(bandsplit_apply_kernel<float>, dim3(B * T), dim3(256), 0, st, spec, (const Band*)bands, stats,
                       gamma, beta, (float*)xnb, B, T, F, K, ldx, eps, (bf16_t*)nullptr);
  URSE_CHECK_LAUNCH("urse_bandsplit_norm_fwd");
  return URSE_OK;
}

extern "C" int urse_bandsplit_norm_bwd(const float* spec, const float* dxnb, const int32_t* bands, const double* stats,
                                       float* dgamma, float* dbeta, int B, int T, int F, int K, int ldx, float eps,
                                       void* stream) {
  URSE_CHECK_ARG(spec && dxnb && bands && stats && dgamma && dbeta, "urse_bandsplit_norm_bwd: null pointer");
  const int tchunk = 64;
  hipLaunchKernelGGL(bandsplit_bwd_affine_kernel, dim3(K, B, ceil_div(T, tchunk)), dim3(128), 0, (hipStream_t)stream,
                     spec, dxnb, (const Band*)bands, stats, dgamma, dbeta, T, F, ldx, eps, tchunk);
  URSE_CHECK_LAUNCH("urse_bandsplit_norm_bwd");
  return URSE_OK;
}

extern "C" int urse_glu_mask_apply_fwd(const float* pre_m, const float* pre_r, const float* x, float* out,
                                       const int32_t* bands, const int32_t* f2k, int64_t rows, int F, int ldp,
                                       void* stream) {
  URSE_CHECK_ARG(pre_m && pre_r && x && out && bands && f2k && rows > 0 && F > 0, "urse_glu_mask_apply_fwd: bad argument");
  hipLaunchKernelGGL(glu_mask_apply_kernel, dim3(grid_for(rows * F)), dim3(256), 0, (hipStream_t)stream, pre_m, pre_r,
                     (const float2*)x, (float2*)out, (const Band*)bands, f2k, (long)rows, F, ldp);
  URSE_CHECK_LAUNCH("urse_glu_mask_apply_fwd");
  return URSE_OK;
}

extern "C" int urse_glu_mask_apply_bwd(const float* pre_m, const float* pre_r, const float* x, const float* dout,
                                       void* dpre_m, void* dpre_r, const int32_t* bands, const int32_t* f2k,
                                       int64_t rows, int F, int ldp, int out_dtype, void* stream) {
  URSE_CHECK_ARG(pre_m && pre_r && x && dout && dpre_m && dpre_r && bands && f2k, "urse_glu_mask_apply_bwd: null pointer");
  dim3 g(grid_for(rows * F)), b(256);
  if (out_dtype == URSE_BF16)
    hipLaunchKernelGGL(glu_mask_apply_bwd_kernel<bf16_t>, g, b, 0, (hipStream_t)stream, pre_m, pre_r, (const float2*)x,
                       (const float2*)dout, (bf16_t*)dpre_m, (bf16_t*)dpre_r, (const Band*)bands, f2k, (long)rows, F, ldp);
  else
    hipLaunchKernelGGL(glu_mask_apply_bwd_kernel<float>, g, b, 0, (hipStream_t)stream, pre_m, pre_r, (const float2*)x,
                       (const float2*)dout, (float*)dpre_m, (float*)dpre_r, (const Band*)bands, f2k, (long)rows, F, ldp);
  URSE_CHECK_LAUNCH("urse_glu_mask_apply_bwd");
  return URSE_OK;
}

extern "C" int urse_axpby(const float* x, float* y, float a, float b, int64_t n, void* stream) {
  URSE_CHECK_ARG(x && y && n > 0, "urse_axpby: bad argument");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, (long)n);
  URSE_CHECK_LAUNCH("urse_axpby");
  return URSE_OK;
}

// ---- torch.nan_to_num(x, nan=0): NaN -> 0, +-inf -> +-FLT_MAX  (flow_model.py:156-157), and x <- x * s[0] (device scalar)
namespace urse {
__global__ void __launch_bounds__(256) nan_to_num_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (isnan(v)) v = 0.f;
    else if (isinf(v)) v = v > 0.f ? 3.402823466e+38f : -3.402823466e+38f;
    y[i] = v;
  }
}
__global__ void __launch_bounds__(256) scale_by_kernel(float* __restrict__ x, const float* __restrict__ s, long n) {
  const float a = *s;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= a;
}
}  // namespace urse
extern "C" int urse_nan_to_num(const float* x, float* y, int64_t n, void* stream) {
  URSE_CHECK_ARG(x && y && n > 0, "urse_nan_to_num: bad argument");
  hipLaunchKernelGGL(nan_to_num_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
  URSE_CHECK_LAUNCH("urse_nan_to_num");
  return URSE_OK;
}
extern "C" int urse_scale_by_device_scalar(float* x, const float* s, int64_t n, void* stream) {
  URSE_CHECK_ARG(x && s && n > 0, "urse_scale_by_device_scalar: bad argument");
  hipLaunchKernelGGL(scale_by_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, s, (long)n);
  URSE_CHECK_LAUNCH("urse_scale_by_device_scalar");
  return URSE_OK;
}

// ---- peak normalisation: x <- x / max|x| * peak   (inference.py:60) -------------------------------------
namespace urse {
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ out, long n) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order like uints
}
__global__ void __launch_bounds__(256) scale_by_peak_kernel(float* __restrict__ x, const unsigned* __restrict__ mx,
                                                            float peak, long n) {
  const float s = peak / __uint_as_float(*mx);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= s;
}
}  // namespace urse

extern "C" int urse_peak_normalize(float* x, int64_t n, float peak, void* scratch, void* stream) {
  URSE_CHECK_ARG(x && scratch && n > 0, "urse_peak_normalize: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(scratch, 0, 4, st);
  hipLaunchKernelGGL(urse::absmax_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, (unsigned*)scratch, (long)n);
  hipLaunchKernelGGL(urse::scale_by_peak_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, (const unsigned*)scratch, peak,
                     (long)n);
  URSE_CHECK_LAUNCH("urse_peak_normalize");
  return URSE_OK;
}
