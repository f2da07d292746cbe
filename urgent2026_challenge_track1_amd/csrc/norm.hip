// GroupNorm(1, C) statistics / apply / backward for the channel-last activation layout, plus the
// small packing kernels (cast + transpose + zero-pad) that feed the MFMA GEMMs.
//
// Activation layout: x f32 [B, T, Kg, W]; a "group" is (b, kg) and covers T rows of W contiguous
// values (row pitch Kg*W).  The dual-path norms use Kg = 1, W = K*N (per-utterance statistics over
// (N, T, K), espnet choose_norm("GN") == nn.GroupNorm(1, N) on [B,N,T,K]); the mask-decoder norms use
// Kg = K, W = N (nn.GroupNorm(1, N) on [B,N,T] per band).  gamma/beta are indexed by
// kg * gstride + (col % N).  HBM-bound: statistics are reduced with wave shuffles + f64 atomics,
// the apply pass reads x once and writes the (bf16|f32) zero-padded GEMM operand once.
// Reference twin: baseline_code/models/bsrnn_flowse.py:291,302 (norm_time / norm_freq), :119-136.
#include "urse_common.h"

namespace urse {

struct GnShape {
  int B, T, Kg, W, N, Np;  // Np = padded channel count of the output rows
  int gstride;             // gamma/beta stride between kg groups (0 = shared)
};

__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats,
                                                       GnShape s, int rows_per_block) {
  __shared__ double red[8];
  const int b = blockIdx.z, kg = blockIdx.y;
  const int t0 = blockIdx.x * rows_per_block;
  int t1 = t0 + rows_per_block;
  if (t1 > s.T) t1 = s.T;
  const long pitch = (long)s.Kg * s.W;
  const float* base = x + ((long)b * s.T) * pitch + (long)kg * s.W;
  const int w4 = s.W >> 2;
  double sum = 0.0, sq = 0.0;
  for (int t = t0; t < t1; ++t) {
    const float4* row = reinterpret_cast<const float4*>(base + (long)t * pitch);
    float ls = 0.f, lq = 0.f;
    for (int i = threadIdx.x; i < w4; i += blockDim.x) {
      const float4 v = row[i];
      ls += (v.x + v.y) + (v.z + v.w);
      lq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    sum += ls;
    sq += lq;
  }
  sum = wave_sum_d(sum);
  sq = wave_sum_d(sq);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = sum; red[4 + w] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = red[0] + red[1] + red[2] + red[3], q = red[4] + red[5] + red[6] + red[7];
    atomicAdd(stats + ((long)b * s.Kg + kg) * 2, a);
    atomicAdd(stats + ((long)b * s.Kg + kg) * 2 + 1, q);
  }
}

__device__ __forceinline__ void gn_mean_rstd(const double* stats, long g, double cnt, float eps, float* mean,
                                             float* rstd) {
  const double m = stats[g * 2] / cnt;
  double var = stats[g * 2 + 1] / cnt - m * m;
  if (var < 0.0) var = 0.0;
  *mean = (float)m;
  *rstd = (float)(1.0 / sqrt(var + (double)eps));
}

// Index walk shared by the two apply kernels: block <-> (chunk of (t, v) pairs, kg, b), so the group's statistics are uniform
// per block (one f64 evaluation per thread, not per element) and a thread steps through its N-vectors with adds only.
// (Before: a flat index split by four 64-bit divisions, and the f64 mean / rstd re-derived, per element.)
constexpr int GN_ITER = 16;
struct GnWalk {
  int t, v, dt, dv, wn, left;
  __device__ __forceinline__ void init(int tv0, int vpb, int wn_, int ntv) {
    wn = wn_;
    t = tv0 / wn; v = tv0 - t * wn;
    dt = vpb / wn; dv = vpb - dt * wn;
    left = tv0 < ntv ? (ntv - tv0 + vpb - 1) / vpb : 0;
    if (left > GN_ITER) left = GN_ITER;
  }
  __device__ __forceinline__ void next() {
    t += dt; v += dv;
    if (v >= wn) { v -= wn; ++t; }
  }
};

// one thread per 4 channels of one N-vector; output rows are [B*T*Kg*(W/N)][Np]
template <typename TO>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const double* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ add, TO* __restrict__ y, GnShape s,
                                                       float eps, bf16_t* __restrict__ y2 = nullptr) {
  // y2 (f16 forward mode, training): the same rows once more in bf16 - the weight-gradient GEMMs multiply them with bf16 gate gradients
  const int n4 = s.Np >> 2, vpb = 256 / n4, wn = s.W / s.N;
  const int b = blockIdx.z, kg = blockIdx.y;
  const int slot = threadIdx.x / n4, c = (threadIdx.x - slot * n4) * 4;
  if (slot >= vpb) return;
  float mean, rstd;
  gn_mean_rstd(stats, (long)b * s.Kg + kg, (double)s.T * s.W, eps, &mean, &rstd);
  float g[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < s.N) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = gamma[(long)kg * s.gstride + c + j];
      be[j] = beta[(long)kg * s.gstride + c + j];
      ad[j] = add ? add[(long)b * s.N + c + j] : 0.f;   // per-(utterance, channel) additive term (flow time embedding)
    }
  }
  GnWalk wk;
  wk.init(blockIdx.x * (vpb * GN_ITER) + slot, vpb, wn, s.T * wn);
  for (int i = 0; i < wk.left; ++i, wk.next()) {
    const long vec = (((long)b * s.T + wk.t) * s.Kg + kg) * wn + wk.v;
    TO o[4];
    float of[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < s.N) {
      const float4 v = *reinterpret_cast<const float4*>(x + vec * s.N + c);
      of[0] = (v.x - mean) * rstd * g[0] + be[0] + ad[0];      // (same arithmetic as ever: results unchanged)
      of[1] = (v.y - mean) * rstd * g[1] + be[1] + ad[1];
      of[2] = (v.z - mean) * rstd * g[2] + be[2] + ad[2];
      of[3] = (v.w - mean) * rstd * g[3] + be[3] + ad[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = from_f32<TO>(of[j]);
    TO* dst = y + vec * s.Np + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = o[j];
    if (y2) {
      uint2 pk;
      pk.x = pack2<bf16_t>(of[0], of[1]);
      pk.y = pack2<bf16_t>(of[2], of[3]);
      *reinterpret_cast<uint2*>(y2 + vec * s.Np + c) = pk;
    }
  }
}

// backward pass 1: per-group s1 = sum dy*gamma, s2 = sum dy*gamma*xhat (f64 atomics), per-channel
// dgamma += sum dy*xhat, dbeta += sum dy (f32 atomics).  thread <-> channel, block <-> (row chunk, kg, b)
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const double* __restrict__ stats,
                                                            const float* __restrict__ gamma, double* __restrict__ sums,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            GnShape s, float eps, int rows_per_block) {
  __shared__ double red[8];
  const int b = blockIdx.z, kg = blockIdx.y;
  const int t0 = blockIdx.x * rows_per_block;
  int t1 = t0 + rows_per_block;
  if (t1 > s.T) t1 = s.T;
  const long pitch = (long)s.Kg * s.W;
  const long base = ((long)b * s.T) * pitch + (long)kg * s.W;
  float mean, rstd;
  gn_mean_rstd(stats, (long)b * s.Kg + kg, (double)s.T * s.W, eps, &mean, &rstd);
  double s1 = 0.0, s2 = 0.0;
  // each thread owns channels n = tid, tid + 256, ... (N <= 256 in practice) across all W/N vectors of a row
  for (int n = threadIdx.x; n < s.N; n += blockDim.x) {
    const float g = gamma[(long)kg * s.gstride + n];
    float dg = 0.f, db = 0.f, a1 = 0.f, a2 = 0.f;
    for (int t = t0; t < t1; ++t) {
      for (int v = 0; v < s.W; v += s.N) {
        const long off = base + (long)t * pitch + v + n;
        const float xh = (x[off] - mean) * rstd;
        const float d = dy[off];
        dg += d * xh;
        db += d;
        a1 += d * g;
        a2 += d * g * xh;
      }
    }
    atomicAdd(dgamma + (long)kg * s.gstride + n, dg);
    atomicAdd(dbeta + (long)kg * s.gstride + n, db);
    s1 += a1;
    s2 += a2;
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = s1; red[4 + w] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums + ((long)b * s.Kg + kg) * 2, red[0] + red[1] + red[2] + red[3]);
    atomicAdd(sums + ((long)b * s.Kg + kg) * 2 + 1, red[4] + red[5] + red[6] + red[7]);
  }
}

// backward pass 1 for N % 4 == 0: the same sums with one thread per 4 channels of one N-vector (16-byte loads, 256 / (N/4)
// vectors in flight per pass), per-channel partials of the vector slots combined through LDS before the atomics
__global__ void __launch_bounds__(256) gn_bwd_reduce4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const double* __restrict__ stats,
                                                             const float* __restrict__ gamma, double* __restrict__ sums,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             GnShape s, float eps, int rows_per_block) {
  __shared__ float4 red_g[256], red_b[256];
  __shared__ double red[8];
  const int n4 = s.N >> 2, vpb = 256 / n4, wn = s.W / s.N;
  const int b = blockIdx.z, kg = blockIdx.y;
  const int slot = threadIdx.x / n4, c = (threadIdx.x - slot * n4) * 4;
  const int t0 = blockIdx.x * rows_per_block;
  int t1 = t0 + rows_per_block;
  if (t1 > s.T) t1 = s.T;
  float mean, rstd;
  gn_mean_rstd(stats, (long)b * s.Kg + kg, (double)s.T * s.W, eps, &mean, &rstd);
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
  float a1 = 0.f, a2 = 0.f;
  if (slot < vpb) {
    float ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ga[j] = gamma[(long)kg * s.gstride + c + j];
    const int nvec = (t1 - t0) * wn;                   // N-vectors of this block's rows
    int t = t0 + slot / wn, v = slot - (slot / wn) * wn;
    const int dt = vpb / wn, dv = vpb - dt * wn;
    // four vectors in flight per thread: offsets first, then the eight 16-byte loads, then the arithmetic
    const long rowp = (long)s.Kg * s.W;                                  // elements per (b, t) row
    const long gbase = ((long)b * s.T * s.Kg + kg) * (long)s.W + c;      // (b, t = 0, kg, v = 0, c)
    auto acc = [&](const float4& xv, const float4& dv4) {
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (xs[q] - mean) * rstd;
        dg[q] += ds[q] * xh;
        db[q] += ds[q];
        a1 += ds[q] * ga[q];
        a2 += ds[q] * ga[q] * xh;
      }
    };
    int j = slot;
    for (; j + 3 * vpb < nvec; j += 4 * vpb) {
      long off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        off[u] = gbase + (long)t * rowp + v * s.N;
        t += dt; v += dv;
        if (v >= wn) { v -= wn; ++t; }
      }
      float4 xv[4], dv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = *reinterpret_cast<const float4*>(x + off[u]);
        dv4[u] = *reinterpret_cast<const float4*>(dy + off[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc(xv[u], dv4[u]);
    }
    for (; j < nvec; j += vpb) {
      const long off = gbase + (long)t * rowp + v * s.N;
      acc(*reinterpret_cast<const float4*>(x + off), *reinterpret_cast<const float4*>(dy + off));
      t += dt; v += dv;
      if (v >= wn) { v -= wn; ++t; }
    }
  }
  red_g[threadIdx.x] = make_float4(dg[0], dg[1], dg[2], dg[3]);
  red_b[threadIdx.x] = make_float4(db[0], db[1], db[2], db[3]);
  double s1 = wave_sum_d((double)a1), s2 = wave_sum_d((double)a2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[w] = s1; red[4 + w] = s2; }
  __syncthreads();
  if (threadIdx.x < n4) {
    float4 g = red_g[threadIdx.x], bb = red_b[threadIdx.x];
    for (int q = 1; q < vpb; ++q) {
      const float4 g2 = red_g[threadIdx.x + q * n4], b2 = red_b[threadIdx.x + q * n4];
      g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
      bb.x += b2.x; bb.y += b2.y; bb.z += b2.z; bb.w += b2.w;
    }
    float* pg = dgamma + (long)kg * s.gstride + c;
    float* pb = dbeta + (long)kg * s.gstride + c;
    atomicAdd(pg, g.x); atomicAdd(pg + 1, g.y); atomicAdd(pg + 2, g.z); atomicAdd(pg + 3, g.w);
    atomicAdd(pb, bb.x); atomicAdd(pb + 1, bb.y); atomicAdd(pb + 2, bb.z); atomicAdd(pb + 3, bb.w);
  }
  if (threadIdx.x == 0) {
    atomicAdd(sums + ((long)b * s.Kg + kg) * 2, red[0] + red[1] + red[2] + red[3]);
    atomicAdd(sums + ((long)b * s.Kg + kg) * 2 + 1, red[4] + red[5] + red[6] + red[7]);
  }
}

// backward pass 2: dx = rstd * (gamma*dy - s1/n - xhat*s2/n) (+ dres), float4 per thread
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const double* __restrict__ stats,
                                                           const double* __restrict__ sums,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ dres, float* __restrict__ dx,
                                                           GnShape s, float eps, bf16_t* __restrict__ dxp, int ldp) {
  const int n4 = s.N >> 2, vpb = 256 / n4, wn = s.W / s.N;
  const int b = blockIdx.z, kg = blockIdx.y;
  const int slot = threadIdx.x / n4, c = (threadIdx.x - slot * n4) * 4;
  if (slot >= vpb) return;
  const long g = (long)b * s.Kg + kg;
  const double cnt = (double)s.T * s.W;
  float mean, rstd;
  gn_mean_rstd(stats, g, cnt, eps, &mean, &rstd);
  const float m1 = (float)(sums[g * 2] / cnt), m2 = (float)(sums[g * 2 + 1] / cnt);
  float ga[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ga[j] = gamma[(long)kg * s.gstride + c + j];
  GnWalk wk;
  wk.init(blockIdx.x * (vpb * GN_ITER) + slot, vpb, wn, s.T * wn);
  for (int i = 0; i < wk.left; ++i, wk.next()) {
    const long vec = (((long)b * s.T + wk.t) * s.Kg + kg) * wn + wk.v;
    const long off = vec * s.N + c;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    const float4 dv = *reinterpret_cast<const float4*>(dy + off);
    float4 o;
    o.x = rstd * (ga[0] * dv.x - m1 - (xv.x - mean) * rstd * m2);
    o.y = rstd * (ga[1] * dv.y - m1 - (xv.y - mean) * rstd * m2);
    o.z = rstd * (ga[2] * dv.z - m1 - (xv.z - mean) * rstd * m2);
    o.w = rstd * (ga[3] * dv.w - m1 - (xv.w - mean) * rstd * m2);
    if (dres) {
      const float4 r = *reinterpret_cast<const float4*>(dres + off);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(dx + off) = o;
    if (dxp) {
      // bf16, K-padded copy of the gradient stream: the A operand of the next half layer's first dgrad GEMM
      bf16_t* pr = dxp + vec * ldp;
      uint2 pk;
      pk.x = (unsigned)f32_to_bf16(o.x) | ((unsigned)f32_to_bf16(o.y) << 16);
      pk.y = (unsigned)f32_to_bf16(o.z) | ((unsigned)f32_to_bf16(o.w) << 16);
      *reinterpret_cast<uint2*>(pr + c) = pk;
      if (c + 4 >= s.N)
        for (int z = s.N; z < ldp; z += 4) *reinterpret_cast<uint2*>(pr + z) = make_uint2(0u, 0u);
    }
  }
}

// out[r, c] (TO, pitch ldo, zero outside [rows, cols]) = in[r, c] or in[c, r] (f32|bf16 source)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) pack2d_kernel(const TI* __restrict__ in, long ldi, TO* __restrict__ out,
                                                     long ldo, int rows, int cols, int out_rows, int out_cols,
                                                     int transpose) {
  const long total = (long)out_rows * out_cols;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / out_cols), c = (int)(idx - (long)r * out_cols);
    float v = 0.f;
    if (r < rows && c < cols) v = to_f32<TI>(transpose ? in[(long)c * ldi + r] : in[(long)r * ldi + c]);
    out[(long)r * ldo + c] = from_f32<TO>(v);
  }
}

}  // namespace urse

using namespace urse;

static int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

static int make_shape(GnShape* s, int B, int T, int Kg, int W, int N, int Np, int gstride, const char* who) {
  URSE_CHECK_ARG(B > 0 && T > 0 && Kg > 0 && W > 0 && N > 0 && W % N == 0 && N % 4 == 0 && Np % 4 == 0 && Np >= N,
                 "%s: bad shape B%d T%d Kg%d W%d N%d Np%d", who, B, T, Kg, W, N, Np);
  s->B = B; s->T = T; s->Kg = Kg; s->W = W; s->N = N; s->Np = Np; s->gstride = gstride;
  return URSE_OK;
}

static int launch_gn_apply(dim3 grid_a, hipStream_t st, const float* x, const double* stats, const float* gamma, const float* beta,
                           const float* add, void* y, void* y_bf16, const GnShape& s, float eps, int out_dtype, const char* who) {
  URSE_CHECK_ARG(!y_bf16 || (out_dtype == URSE_F16 && ((uintptr_t)y_bf16 % 8) == 0), "%s: the bf16 copy goes with f16 output only", who);
  if (out_dtype == URSE_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid_a, dim3(256), 0, st, x, stats, gamma, beta, add, (bf16_t*)y, s, eps, (bf16_t*)nullptr);
  else if (out_dtype == URSE_F16)
    hipLaunchKernelGGL(gn_apply_kernel<f16_t>, grid_a, dim3(256), 0, st, x, stats, gamma, beta, add, (f16_t*)y, s, eps, (bf16_t*)y_bf16);
  else if (out_dtype == URSE_F32)
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid_a, dim3(256), 0, st, x, stats, gamma, beta, add, (float*)y, s, eps, (bf16_t*)nullptr);
  else { set_error("%s: bad output dtype %d", who, out_dtype); return URSE_ERR_INVALID_ARG; }
  return URSE_OK;
}

extern "C" int urse_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* add, void* y,
                                  double* stats,
                                  int B, int T, int Kg, int W, int N, int Np, int gstride, float eps, int out_dtype,
                                  void* y_bf16, void* stream) {
  GnShape s;
  int rc = make_shape(&s, B, T, Kg, W, N, Np, gstride, "urse_groupnorm_fwd");
  if (rc) return rc;
  URSE_CHECK_ARG(x && gamma && beta && y && stats, "urse_groupnorm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(stats, 0, sizeof(double) * 2 * B * Kg, st);
  int nblk = ceil_div((long)T * W, 32768);  // ~32k values per block
  if (nblk > T) nblk = T;
  const int rpb = ceil_div(T, nblk);
  dim3 grid(ceil_div(T, rpb), Kg, B);
  hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, st, x, stats, s, rpb);
  URSE_CHECK_ARG(Np / 4 <= 256, "urse_groupnorm_fwd: Np %d too wide", Np);
  const int vpb_f = 256 / (Np / 4);
  dim3 grid_a(ceil_div((long)T * (W / N), (long)vpb_f * GN_ITER), Kg, B);
  rc = launch_gn_apply(grid_a, st, x, stats, gamma, beta, add, y, y_bf16, s, eps, out_dtype, "urse_groupnorm_fwd");
  if (rc) return rc;
  URSE_CHECK_LAUNCH("urse_groupnorm_fwd");
  return URSE_OK;
}

// the two halves of urse_groupnorm_fwd on their own: statistics only (accumulates into `stats`, which the caller has zeroed),
// and normalisation with statistics that exist already (e.g. from urse_gemm_nt_gnstats)
extern "C" int urse_groupnorm_stats(const float* x, double* stats, int B, int T, int Kg, int W, int N, void* stream) {
  GnShape s;
  int rc = make_shape(&s, B, T, Kg, W, N, N, 0, "urse_groupnorm_stats");
  if (rc) return rc;
  URSE_CHECK_ARG(x && stats, "urse_groupnorm_stats: null pointer");
  int nblk = ceil_div((long)T * W, 32768);
  if (nblk > T) nblk = T;
  const int rpb = ceil_div(T, nblk);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(ceil_div(T, rpb), Kg, B), dim3(256), 0, (hipStream_t)stream, x, stats, s, rpb);
  URSE_CHECK_LAUNCH("urse_groupnorm_stats");
  return URSE_OK;
}

extern "C" int urse_groupnorm_apply(const float* x, const float* gamma, const float* beta, const float* add, void* y,
                                    const double* stats, int B, int T, int Kg, int W, int N, int Np, int gstride, float eps,
                                    int out_dtype, void* y_bf16, void* stream) {
  GnShape s;
  int rc = make_shape(&s, B, T, Kg, W, N, Np, gstride, "urse_groupnorm_apply");
  if (rc) return rc;
  URSE_CHECK_ARG(x && gamma && beta && y && stats && Np / 4 <= 256, "urse_groupnorm_apply: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int vpb_f = 256 / (Np / 4);
  dim3 grid_a(ceil_div((long)T * (W / N), (long)vpb_f * GN_ITER), Kg, B);
  rc = launch_gn_apply(grid_a, st, x, stats, gamma, beta, add, y, y_bf16, s, eps, out_dtype, "urse_groupnorm_apply");
  if (rc) return rc;
  URSE_CHECK_LAUNCH("urse_groupnorm_apply");
  return URSE_OK;
}

extern "C" int urse_groupnorm_bwd(const float* x, const float* dy, const double* stats, const float* gamma,
                                  const float* dres, float* dx, float* dgamma, float* dbeta, double* sums, int B,
                                  int T, int Kg, int W, int N, int gstride, float eps, void* dx_packed, int ldp,
                                  void* stream) {
  URSE_CHECK_ARG(!dx_packed || (ldp >= N && ldp % 4 == 0 && N % 4 == 0 && ((uintptr_t)dx_packed % 8) == 0),
                 "urse_groupnorm_bwd: packed copy needs N, ldp multiples of 4 and an 8-byte aligned buffer");
  GnShape s;
  int rc = make_shape(&s, B, T, Kg, W, N, N, gstride, "urse_groupnorm_bwd");
  if (rc) return rc;
  URSE_CHECK_ARG(x && dy && stats && gamma && dx && dgamma && dbeta && sums, "urse_groupnorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(sums, 0, sizeof(double) * 2 * B * Kg, st);
  // elements per workgroup of pass 1.  Every workgroup ends with 2 N f32 atomics on the SAME 2 N addresses (dgamma, dbeta): at
  // 32768 elements (2,592 workgroups at C2) the kernel was bound by that serialisation, not by its 684 MB of reads
  // (same-box A/B of the step: 169.8 ms at 32768, 168.6 at 65536, 167.3 at 131072, 167.1 at 262144)
  static const long red_elems = getenv("URSE_GN_REDUCE_ELEMS") ? atol(getenv("URSE_GN_REDUCE_ELEMS")) : 262144;
  int nblk = ceil_div((long)T * W, red_elems);
  if (nblk > T) nblk = T;
  const int rpb = ceil_div(T, nblk);
  dim3 grid(ceil_div(T, rpb), Kg, B);
  static const bool reduce1 = getenv("URSE_GN_REDUCE1") != nullptr;
  if (N % 4 == 0 && N / 4 <= 256 && W % N == 0 && !reduce1)
    hipLaunchKernelGGL(gn_bwd_reduce4_kernel, grid, dim3(256), 0, st, x, dy, stats, gamma, sums, dgamma, dbeta, s, eps, rpb);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, grid, dim3(256), 0, st, x, dy, stats, gamma, sums, dgamma, dbeta, s, eps,
                       rpb);
  URSE_CHECK_ARG(N / 4 <= 256, "urse_groupnorm_bwd: N %d too wide", N);
  const int vpb_b = 256 / (N / 4);
  dim3 grid_a(ceil_div((long)T * (W / N), (long)vpb_b * GN_ITER), Kg, B);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid_a, dim3(256), 0, st, x, dy, stats, sums, gamma, dres, dx, s, eps,
                     (bf16_t*)dx_packed, ldp);
  URSE_CHECK_LAUNCH("urse_groupnorm_bwd");
  return URSE_OK;
}

// the apply pass of urse_groupnorm_bwd alone, for sums that came out of the GEMM that produced dy (urse_gemm_nt_gnbwd)
extern "C" int urse_groupnorm_bwd_apply(const float* x, const float* dy, const double* stats, const double* sums, const float* gamma,
                                        const float* dres, float* dx, int B, int T, int Kg, int W, int N, int gstride, float eps,
                                        void* dx_packed, int ldp, void* stream) {
  URSE_CHECK_ARG(!dx_packed || (ldp >= N && ldp % 4 == 0 && N % 4 == 0 && ((uintptr_t)dx_packed % 8) == 0),
                 "urse_groupnorm_bwd_apply: packed copy needs N, ldp multiples of 4 and an 8-byte aligned buffer");
  GnShape s;
  int rc = make_shape(&s, B, T, Kg, W, N, N, gstride, "urse_groupnorm_bwd_apply");
  if (rc) return rc;
  URSE_CHECK_ARG(x && dy && stats && gamma && dx && sums && N % 4 == 0 && N / 4 <= 256 && W % N == 0, "urse_groupnorm_bwd_apply: bad argument");
  const int vpb_b = 256 / (N / 4);
  dim3 grid_a(ceil_div((long)T * (W / N), (long)vpb_b * GN_ITER), Kg, B);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid_a, dim3(256), 0, (hipStream_t)stream, x, dy, stats, sums, gamma, dres, dx, s, eps,
                     (bf16_t*)dx_packed, ldp);
  URSE_CHECK_LAUNCH("urse_groupnorm_bwd_apply");
  return URSE_OK;
}

extern "C" int urse_pack2d(const void* in, int64_t ldi, int in_dtype, void* out, int64_t ldo, int out_dtype, int rows,
                           int cols, int out_rows, int out_cols, int transpose, void* stream) {
  URSE_CHECK_ARG(in && out && rows >= 0 && cols >= 0 && out_rows > 0 && out_cols > 0 && ldo >= out_cols,
                 "urse_pack2d: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)out_rows * out_cols;
  dim3 g(grid_for(total)), b(256);
  if (in_dtype == URSE_F32 && out_dtype == URSE_BF16)
    hipLaunchKernelGGL((pack2d_kernel<float, bf16_t>), g, b, 0, st, (const float*)in, (long)ldi, (bf16_t*)out,
                       (long)ldo, rows, cols, out_rows, out_cols, transpose);
  else if (in_dtype == URSE_F32 && out_dtype == URSE_F32)
    hipLaunchKernelGGL((pack2d_kernel<float, float>), g, b, 0, st, (const float*)in, (long)ldi, (float*)out, (long)ldo,
                       rows, cols, out_rows, out_cols, transpose);
  else if (in_dtype == URSE_BF16 && out_dtype == URSE_BF16)
    hipLaunchKernelGGL((pack2d_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)in, (long)ldi, (bf16_t*)out,
                       (long)ldo, rows, cols, out_rows, out_cols, transpose);
  else if (in_dtype == URSE_BF16 && out_dtype == URSE_F32)
    hipLaunchKernelGGL((pack2d_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)in, (long)ldi, (float*)out,
                       (long)ldo, rows, cols, out_rows, out_cols, transpose);
  else if (in_dtype == URSE_F32 && out_dtype == URSE_F16)
    hipLaunchKernelGGL((pack2d_kernel<float, f16_t>), g, b, 0, st, (const float*)in, (long)ldi, (f16_t*)out,
                       (long)ldo, rows, cols, out_rows, out_cols, transpose);
  else if (in_dtype == URSE_F16 && out_dtype == URSE_F32)
    hipLaunchKernelGGL((pack2d_kernel<f16_t, float>), g, b, 0, st, (const f16_t*)in, (long)ldi, (float*)out,
                       (long)ldo, rows, cols, out_rows, out_cols, transpose);
  else { set_error("urse_pack2d: bad dtype"); return URSE_ERR_INVALID_ARG; }
  URSE_CHECK_LAUNCH("urse_pack2d");
  return URSE_OK;
}

// ---- segmented pack: S independent (cast + optional transpose + zero pad) copies in one launch -------
namespace urse {
struct PackSeg {  // int64 x 8, built by the host once per model / dtype
  long in_off, in_rows, in_cols, in_ld;      // source block (elements, relative to `in`)
  long out_off, out_rows, out_cols, out_ld;  // destination block (elements, relative to `out`)
};
template <typename TO>
__global__ void __launch_bounds__(256) pack_seg_kernel(const float* __restrict__ in, TO* __restrict__ out,
                                                       const PackSeg* __restrict__ segs, int transpose) {
  const PackSeg s = segs[blockIdx.y];
  const long total = s.out_rows * s.out_cols;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / s.out_cols, c = idx - r * s.out_cols;
    float v = 0.f;
    if (!transpose) {
      if (r < s.in_rows && c < s.in_cols) v = in[s.in_off + r * s.in_ld + c];
    } else {
      if (c < s.in_rows && r < s.in_cols) v = in[s.in_off + c * s.in_ld + r];
    }
    out[s.out_off + r * s.out_ld + c] = from_f32<TO>(v);
  }
}
}  // namespace urse

extern "C" int urse_pack_segments(const float* in, void* out, const void* segs, int nseg, int blocks_per_seg,
                                  int transpose, int out_dtype, void* stream) {
  URSE_CHECK_ARG(in && out && segs && nseg > 0 && blocks_per_seg > 0, "urse_pack_segments: bad argument");
  dim3 g(blocks_per_seg, nseg), b(256);
  if (out_dtype == URSE_BF16)
    hipLaunchKernelGGL(urse::pack_seg_kernel<urse::bf16_t>, g, b, 0, (hipStream_t)stream, in, (urse::bf16_t*)out,
                       (const urse::PackSeg*)segs, transpose);
  else if (out_dtype == URSE_F16)
    hipLaunchKernelGGL(urse::pack_seg_kernel<urse::f16_t>, g, b, 0, (hipStream_t)stream, in, (urse::f16_t*)out,
                       (const urse::PackSeg*)segs, transpose);
  else
    hipLaunchKernelGGL(urse::pack_seg_kernel<float>, g, b, 0, (hipStream_t)stream, in, (float*)out,
                       (const urse::PackSeg*)segs, transpose);
  URSE_CHECK_LAUNCH("urse_pack_segments");
  return URSE_OK;
}
