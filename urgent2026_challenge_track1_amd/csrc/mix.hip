// On-device dynamic mixing: the numpy / scipy DSP subset of the reference's on-the-fly simulator,
// batched over utterances ([B, L] f32 signals, per-utterance lengths), f64 accumulation where the reference is f64.
//
//   nonsilence_power  espnet2 detect_non_silence (boxcar, 1024 / 512, threshold 0.01; restated from SURVEY A.5) +
//                     the masked mean power used by mix_noise (simulation/simulate_data_from_param.py:121-122)
//   mix_noise         noise aligned to the speech length (wrap-pad or crop at an offset drawn by the host, :108-119),
//                     scale = 10^(-snr/20) * sqrt(Ps) / sqrt(max(Pn, 1e-10)), noisy = speech + scale*noise (:123-126)
//   fir_full          scipy.signal.convolve(x, taps, "full")[:, :L] (add_reverberation :220-230; high-pass :29-56,461)
//                     as a direct f64-accumulated convolution: HBM-light, VALU-bound, no FFT plan per RIR length
//   quantile_clip     np.quantile (linear) + np.clip (:255-276) with an exact 3-pass radix select per utterance
//   zero_segments     packet_loss (:333-341)
//   joint_peak_scale  final 0.9 / max(|noisy|, |speech|, |noise|, 1e-6) normalisation (:576-584)
// All kernels are bandwidth-trivial next to the model (a few MB per utterance); they exist so that rank-local mixing for
// the DP shards needs no host round trip.
#include <math.h>

#include "urse_common.h"

namespace urse {

constexpr int HOP = 512, FRAME = 1024;

// per-hop sums of squares, f64: S[b, h] = sum x[b, 512h .. 512h+511]^2 (samples >= len are zero)
__global__ void __launch_bounds__(256) hop_sumsq_kernel(const float* __restrict__ x, const int* __restrict__ lens, long ld,
                                                        double* __restrict__ S, int nhop) {
  const int b = blockIdx.y, h = blockIdx.x;
  const int len = lens[b];
  const float* xb = x + (long)b * ld;
  double acc = 0.0;
  for (int i = threadIdx.x; i < HOP; i += 256) {
    const long p = (long)h * HOP + i;
    if (p < len) { const double v = xb[p]; acc += v * v; }
  }
  acc = wave_sum_d(acc);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) S[(long)b * nhop + h] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup per utterance: frame powers from hop sums, detect flags, masked mean power
__global__ void __launch_bounds__(256) nonsilence_power_kernel(const double* __restrict__ S, const int* __restrict__ lens,
                                                               int nhop, double threshold, double* __restrict__ power) {
  const int b = blockIdx.x;
  const int len = lens[b];
  const double* Sb = S + (long)b * nhop;
  __shared__ double red[256];
  __shared__ double s_mean;
  const int hops = (len + HOP - 1) / HOP;
  if (len < FRAME) {                       // all samples count
    double acc = 0.0;
    for (int h = threadIdx.x; h < hops; h += 256) acc += Sb[h];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) power[b] = len > 0 ? red[0] / len : 0.0;
    return;
  }
  // padded framing: nadd = (-(len - FRAME) % HOP) % FRAME, T = (len + nadd - FRAME) / HOP + 1
  const int rem = (len - FRAME) % HOP;
  const int nadd = ((rem ? HOP - rem : 0)) % FRAME;
  const int T = (len + nadd - FRAME) / HOP + 1;
  auto hop = [&](int h) -> double { return h < hops ? Sb[h] : 0.0; };
  double acc = 0.0;
  for (int f = threadIdx.x; f < T; f += 256) acc += (hop(f) + hop(f + 1)) / FRAME;
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) s_mean = red[0] / T;
  __syncthreads();
  const double mean_power = s_mean;
  // sample i carries the flag of frame min(i / HOP, T - 1): accumulate whole hops
  double num = 0.0, cnt = 0.0;
  for (int h = threadIdx.x; h < hops; h += 256) {
    const int f = h < T ? h : T - 1;
    const bool on = mean_power == 0.0 ? true : ((hop(f) + hop(f + 1)) / FRAME) / mean_power > threshold;
    if (on) {
      num += Sb[h];
      const long lo = (long)h * HOP;
      cnt += (double)((lo + HOP <= len ? HOP : len - lo));
    }
  }
  __syncthreads();
  red[threadIdx.x] = num;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const double tnum = red[0];
  __syncthreads();
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) power[b] = red[0] > 0.0 ? tnum / red[0] : nan("");   // numpy: mean of an empty selection
}

// noise_al[b, i] = noise[b, wrap/crop(i)]
__global__ void __launch_bounds__(256) noise_align_kernel(const float* __restrict__ noise, const int* __restrict__ nlens, long ldn,
                                                          const int* __restrict__ lens, const int* __restrict__ offsets,
                                                          float* __restrict__ out, long ldo) {
  const int b = blockIdx.y;
  const int len = lens[b], nl = nlens[b], off = offsets[b];
  const float* nb = noise + (long)b * ldn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ldo; i += (long)gridDim.x * 256) {
    float v = 0.f;
    if (i < len && nl > 0) {
      if (nl < len) {                    // np.pad(mode="wrap") with `off` samples in front
        long j = (i - off) % nl;
        if (j < 0) j += nl;
        v = nb[j];
      } else {
        v = nb[(nl > len ? off : 0) + i];
      }
    }
    out[(long)b * ldo + i] = v;
  }
}

__global__ void __launch_bounds__(256) mix_apply_kernel(const float* __restrict__ speech, float* __restrict__ noise,
                                                        float* __restrict__ noisy, const int* __restrict__ lens, long ld,
                                                        const double* __restrict__ ps, const double* __restrict__ pn,
                                                        const float* __restrict__ snr) {
  const int b = blockIdx.y;
  const int len = lens[b];
  const double pnb = pn[b] > 1e-10 ? pn[b] : 1e-10;
  const double scale = pow(10.0, -(double)snr[b] / 20.0) * sqrt(ps[b]) / sqrt(pnb);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ld; i += (long)gridDim.x * 256) {
    const long o = (long)b * ld + i;
    if (i < len) {
      const double n = scale * (double)noise[o];
      noise[o] = (float)n;
      noisy[o] = (float)((double)speech[o] + n);
    } else {
      noise[o] = 0.f;
      noisy[o] = 0.f;
    }
  }
}

// y[b, i] = sum_k taps[tb, k] * x[b, i - k], 0 <= i < len, accumulated in f64 in ascending k.  1024 outputs per workgroup,
// four consecutive ones per thread; taps go through LDS in chunks of 1024 together with the 2048 input samples they touch.
// Per block of four taps a thread reads ONE new aligned quad of inputs (the other quad of its 7-sample window is the
// previous block's) and one broadcast quad of taps for 16 multiply-adds.
constexpr int FIR_TC = 1024;
constexpr int FIR_OUT = 1024;
__global__ void __launch_bounds__(256) fir_full_kernel(const float* __restrict__ x, const int* __restrict__ lens, long ld,
                                                       const float* __restrict__ taps, const int* __restrict__ ntaps,
                                                       long ldt, int taps_per_utt, float* __restrict__ y, int len_add) {
  constexpr int TC = FIR_TC, NO = FIR_OUT;
  __shared__ __align__(16) float st[TC];
  __shared__ __align__(16) float sx[TC + NO];       // sx[j] = x[i0 - k0 - TC + j]
  const int b = blockIdx.y;
  const int len = lens[b] + len_add;
  const long i0 = (long)blockIdx.x * NO;
  float* yb = y + (long)b * ld;
  if (i0 >= len) {
    for (long i = i0 + threadIdx.x; i < i0 + NO && i < ld; i += 256) yb[i] = 0.f;
    return;
  }
  const int tb = taps_per_utt ? b : 0;
  int nt = ntaps[tb];
  if ((long)nt > i0 + NO) nt = (int)(i0 + NO);             // taps beyond the last output index never contribute
  const float* xb = x + (long)b * ld;
  const float* tp = taps + (long)tb * ldt;
  const int t = threadIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < nt; k0 += TC) {
    const int nk = nt - k0 < TC ? nt - k0 : TC;
    const int nkb = (nk + 3) >> 2;                          // blocks of four taps in this chunk (tail zero-padded)
    __syncthreads();
    for (int k = t; k < 4 * nkb; k += 256) st[k] = (k < nk) ? tp[k0 + k] : 0.f;
    const long xlo = i0 - k0 - TC;
    for (int j = TC - 4 * nkb + t; j < TC + NO; j += 256) {
      const long p = xlo + j;
      sx[j] = (p >= 0 && p < len) ? xb[p] : 0.f;
    }
    __syncthreads();
    // output i0 + 4t + c, tap k0 + 4kb + r reads sx[TC + 4(t - kb) + c - r]
    const float4* q = reinterpret_cast<const float4*>(sx) + (TC / 4 + t);
    const float4* tq = reinterpret_cast<const float4*>(st);
    float4 cf = q[0];
    double w[8];
    w[4] = cf.x; w[5] = cf.y; w[6] = cf.z; w[7] = cf.w;
#pragma unroll 2
    for (int kb = 0; kb < nkb; ++kb) {
      const float4 pf = q[-kb - 1];
      const float4 tf = tq[kb];
      w[0] = pf.x; w[1] = pf.y; w[2] = pf.z; w[3] = pf.w;
      const double tk[4] = {(double)tf.x, (double)tf.y, (double)tf.z, (double)tf.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += tk[r] * w[4 + c - r];
      w[4] = w[0]; w[5] = w[1]; w[6] = w[2]; w[7] = w[3];
    }
  }
  const long o = i0 + 4 * t;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (o + c < ld) yb[o + c] = (o + c < len) ? (float)acc[c] : 0.f;
}

// scipy.signal.filtfilt(b, 1.0, x) for an FIR b (high-pass of simulate_data_from_param.py:461): odd extension by
// padlen = 3 * ntaps on both sides, forward filter started in steady state (lfilter_zi * x_ext[0] == "the input was
// x_ext[0] for ever"), reverse, the same again, reverse, crop.  The steady-state start is realised by prefixing
// P = ntaps - 1 copies of the first sample and dropping the first P outputs of a plain causal convolution.
// stage 0: e[j] = prefix | odd_ext(x);  stage 1: e[j] = prefix | reversed(f[P:])
__global__ void __launch_bounds__(256) filtfilt_stage_kernel(const float* __restrict__ src, const int* __restrict__ lens, long lds_,
                                                             float* __restrict__ dst, long ldd, int P, int padlen, int stage) {
  const int b = blockIdx.y;
  const int len = lens[b];
  const int lext = len + 2 * padlen;
  const float* sb = src + (long)b * lds_;
  auto ext = [&](int m) -> float {             // odd extension of x, 0 <= m < lext
    if (m < padlen) return 2.f * sb[0] - sb[padlen - m];
    if (m < padlen + len) return sb[m - padlen];
    return 2.f * sb[len - 1] - sb[len - 2 - (m - padlen - len)];
  };
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < ldd; j += (long)gridDim.x * 256) {
    float v = 0.f;
    if (j < P + lext) {
      const int m = j < P ? 0 : (int)(j - P);
      v = stage == 0 ? ext(m) : sb[P + lext - 1 - m];
    }
    dst[(long)b * ldd + j] = v;
  }
}

// y[i] = f2[P + lext - 1 - (padlen + i)]
__global__ void __launch_bounds__(256) filtfilt_crop_kernel(const float* __restrict__ f2, const int* __restrict__ lens, long ldf,
                                                            float* __restrict__ y, long ld, int P, int padlen) {
  const int b = blockIdx.y;
  const int len = lens[b];
  const int lext = len + 2 * padlen;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ld; i += (long)gridDim.x * 256)
    y[(long)b * ld + i] = i < len ? f2[(long)b * ldf + P + lext - 1 - (padlen + i)] : 0.f;
}

// order-preserving key of a float
__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// exact order statistics ranks[0..3] (0-based) of x[b, 0:len] by a 3-pass (11 + 11 + 10 bit) radix select, one
// workgroup per utterance; then np.quantile's linear interpolation and np.clip in place
__global__ void __launch_bounds__(1024) quantile_clip_kernel(float* __restrict__ x, const int* __restrict__ lens, long ld,
                                                             const float* __restrict__ qmin, const float* __restrict__ qmax,
                                                             float* __restrict__ bounds) {
  __shared__ unsigned hist[4][2048];
  __shared__ unsigned prefix[4], want[4], sel[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lens[b];
  float* xb = x + (long)b * ld;
  if (len <= 0) return;
  // virtual indices q * (n - 1): floor / ceil neighbours
  double vi[2] = {(double)qmin[b] * (len - 1), (double)qmax[b] * (len - 1)};
  if (tid < 4) {
    const double v = vi[tid >> 1];
    long r = (tid & 1) ? (long)ceil(v) : (long)floor(v);
    if (r < 0) r = 0;
    if (r > len - 1) r = len - 1;
    want[tid] = (unsigned)r;
    prefix[tid] = 0u;
  }
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    for (int i = tid; i < 4 * 2048; i += 1024) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const int sh = shifts[pass], nb = bits[pass];
    const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (sh + nb));
    const unsigned p0 = prefix[0], p1 = prefix[1], p2 = prefix[2], p3 = prefix[3];
    for (int i = tid; i < len; i += 1024) {
      const unsigned k = fkey(xb[i]);
      const unsigned d = (k >> sh) & ((1u << nb) - 1u);
      const unsigned hi = k & himask;
      if (hi == p0) atomicAdd(&hist[0][d], 1u);
      if (hi == p1) atomicAdd(&hist[1][d], 1u);
      if (hi == p2) atomicAdd(&hist[2][d], 1u);
      if (hi == p3) atomicAdd(&hist[3][d], 1u);
    }
    __syncthreads();
    if (tid < 4) {                      // serial scan of <= 2048 bins per rank: negligible
      unsigned r = want[tid], d = 0;
      const int n = 1 << nb;
      for (d = 0; d < (unsigned)n; ++d) {
        const unsigned c = hist[tid][d];
        if (r < c) break;
        r -= c;
      }
      want[tid] = r;
      prefix[tid] |= d << sh;
      sel[tid] = prefix[tid];
    }
    __syncthreads();
  }
  // np.quantile linear method: a + (b - a) * t, with numpy's lerp correction for t >= 0.5
  double qv[2];
  for (int s = 0; s < 2; ++s) {
    const double a = funkey(sel[2 * s]), c = funkey(sel[2 * s + 1]);
    const double t = vi[s] - floor(vi[s]);
    double r = a + (c - a) * t;
    if (t >= 0.5) r = c - (c - a) * (1.0 - t);
    if (t == 0.0) r = a;
    qv[s] = r;
  }
  if (tid == 0) { bounds[2 * b] = (float)qv[0]; bounds[2 * b + 1] = (float)qv[1]; }
  const float lo = (float)qv[0], hi = (float)qv[1];
  for (int i = tid; i < len; i += 1024) xb[i] = fminf(fmaxf(xb[i], lo), hi);
}

__global__ void __launch_bounds__(256) zero_segments_kernel(float* __restrict__ x, long ld, const int* __restrict__ seg, int nseg) {
  // seg = [nseg, 3] {utterance, start, end}
  const int s = blockIdx.x;
  if (s >= nseg) return;
  const int b = seg[3 * s];
  long lo = seg[3 * s + 1], hi = seg[3 * s + 2];
  if (hi > ld) hi = ld;
  for (long i = lo + threadIdx.x; i < hi; i += 256) x[(long)b * ld + i] = 0.f;
}

__global__ void __launch_bounds__(256) absmax3_kernel(const float* __restrict__ a, const float* __restrict__ b2,
                                                      const float* __restrict__ c, long ld, unsigned* __restrict__ peak) {
  const int b = blockIdx.y;
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ld; i += (long)gridDim.x * 256) {
    const long o = (long)b * ld + i;
    m = fmaxf(m, fmaxf(fabsf(a[o]), fmaxf(fabsf(b2[o]), fabsf(c[o]))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(peak + b, __float_as_uint(m));   // non-negative floats order like uints
}

__global__ void __launch_bounds__(256) scale3_kernel(float* __restrict__ a, float* __restrict__ b2, float* __restrict__ c, long ld,
                                                     const unsigned* __restrict__ peak, float target, float floor_) {
  const int b = blockIdx.y;
  const float s = target / fmaxf(__uint_as_float(peak[b]), floor_);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ld; i += (long)gridDim.x * 256) {
    const long o = (long)b * ld + i;
    a[o] *= s; b2[o] *= s; c[o] *= s;
  }
}

}  // namespace urse

using namespace urse;

namespace urse {
// resampy.resample (librosa res_type "kaiser_best" / "kaiser_fast": two of the four resamplers of the bandwidth-limitation
// augmentation, simulate_data_from_param.py:233-252): Smith's band-limited interpolation with a linearly interpolated filter
// table, resampy/interpn.py `_resample_loop` restated.  One thread per output sample, float64 index arithmetic and accumulation
// as numpy's (the truncations int(time), int(index_frac) must come out the same), table + its first differences in float64.
__global__ void __launch_bounds__(256) resample_table_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                             const double* __restrict__ win, const double* __restrict__ delta,
                                                             int nwin, int n_orig, int n_out, double time_increment, double scale,
                                                             int num_table, int index_step) {
  const float* xp = x + (long)blockIdx.y * ldx;
  float* yp = y + (long)blockIdx.y * ldy;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_out; t += gridDim.x * blockDim.x) {
    const double time_register = (double)t * time_increment;
    const int n = (int)time_register;
    double frac = scale * (time_register - (double)n);
    double index_frac = frac * (double)num_table;
    int offset = (int)index_frac;
    double eta = index_frac - (double)offset;
    double acc = 0.0;
    int i_max = (nwin - offset) / index_step;
    if (i_max > n + 1) i_max = n + 1;
    for (int i = 0; i < i_max; ++i) {
      const int q = offset + i * index_step;
      acc += (win[q] + eta * delta[q]) * (double)xp[n - i];
    }
    frac = scale - frac;
    index_frac = frac * (double)num_table;
    offset = (int)index_frac;
    eta = index_frac - (double)offset;
    int k_max = (nwin - offset) / index_step;
    if (k_max > n_orig - n - 1) k_max = n_orig - n - 1;
    for (int k = 0; k < k_max; ++k) {
      const int q = offset + k * index_step;
      acc += (win[q] + eta * delta[q]) * (double)xp[n + k + 1];
    }
    yp[t] = (float)acc;
  }
}
}  // namespace urse

extern "C" int urse_resample_table(const float* x, int64_t ldx, float* y, int64_t ldy, const double* win, const double* delta,
                                   int nwin, int P, int n_orig, int n_out, double time_increment, double scale, int num_table,
                                   int index_step, void* stream) {
  URSE_CHECK_ARG(x && y && win && delta && nwin > 0 && P > 0 && n_orig > 0 && n_out > 0 && ldx >= n_orig && ldy >= n_out &&
                     time_increment > 0.0 && scale > 0.0 && scale <= 1.0 && num_table > 0 && index_step > 0,
                 "urse_resample_table: bad argument");
  URSE_CHECK_ARG((double)(n_out - 1) * time_increment < (double)n_orig, "urse_resample_table: the output grid runs past the input");
  hipLaunchKernelGGL(urse::resample_table_kernel, dim3(ceil_div(n_out, 256), P), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y,
                     (long)ldy, win, delta, nwin, n_orig, n_out, time_increment, scale, num_table, index_step);
  URSE_CHECK_LAUNCH("urse_resample_table");
  return URSE_OK;
}

extern "C" int urse_nonsilence_power(const float* x, const int32_t* lens, int B, int64_t ld, double threshold,
                                     double* hop_scratch, double* power, void* stream) {
  URSE_CHECK_ARG(x && lens && hop_scratch && power && B > 0 && ld > 0, "urse_nonsilence_power: bad argument");
  const int nhop = (int)((ld + HOP - 1) / HOP);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hop_sumsq_kernel, dim3(nhop, B), dim3(256), 0, st, x, lens, (long)ld, hop_scratch, nhop);
  hipLaunchKernelGGL(nonsilence_power_kernel, dim3(B), dim3(256), 0, st, (const double*)hop_scratch, lens, nhop, threshold, power);
  URSE_CHECK_LAUNCH("urse_nonsilence_power");
  return URSE_OK;
}

extern "C" int urse_mix_noise(const float* speech, const float* noise_raw, const int32_t* noise_lens, int64_t ldn,
                              const int32_t* lens, const int32_t* offsets, const float* snr_db, int B, int64_t ld,
                              float* noise_out, float* noisy_out, double* scratch, void* stream) {
  URSE_CHECK_ARG(speech && noise_raw && noise_lens && lens && offsets && snr_db && noise_out && noisy_out && scratch && B > 0 &&
                     ld > 0 && ldn > 0, "urse_mix_noise: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nhop = (int)((ld + HOP - 1) / HOP);
  double* hop = scratch;                          // [B, nhop]
  double* ps = scratch + (long)B * nhop;          // [B]
  double* pn = ps + B;                            // [B]
  const int gx = (int)((ld + 255) / 256 < 1024 ? (ld + 255) / 256 : 1024);
  hipLaunchKernelGGL(noise_align_kernel, dim3(gx, B), dim3(256), 0, st, noise_raw, noise_lens, (long)ldn, lens, offsets,
                     noise_out, (long)ld);
  int rc = urse_nonsilence_power(speech, lens, B, ld, 0.01, hop, ps, stream);
  if (rc) return rc;
  rc = urse_nonsilence_power(noise_out, lens, B, ld, 0.01, hop, pn, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(mix_apply_kernel, dim3(gx, B), dim3(256), 0, st, speech, noise_out, noisy_out, lens, (long)ld,
                     (const double*)ps, (const double*)pn, snr_db);
  URSE_CHECK_LAUNCH("urse_mix_noise");
  return URSE_OK;
}

extern "C" int urse_fir_full(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps,
                             int64_t ldt, int taps_per_utt, float* y, void* stream) {
  URSE_CHECK_ARG(x && lens && taps && ntaps && y && x != y && B > 0 && ld > 0 && ldt > 0, "urse_fir_full: bad argument");
  hipLaunchKernelGGL(fir_full_kernel, dim3((unsigned)((ld + FIR_OUT - 1) / FIR_OUT), B), dim3(256), 0, (hipStream_t)stream, x, lens, (long)ld,
                     taps, ntaps, (long)ldt, taps_per_utt, y, 0);
  URSE_CHECK_LAUNCH("urse_fir_full");
  return URSE_OK;
}

extern "C" int urse_filtfilt_fir(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps_dev,
                                 int ntaps, float* y, float* scratch, int64_t lds_, void* stream) {
  URSE_CHECK_ARG(x && lens && taps && ntaps_dev && y && scratch && B > 0 && ld > 0 && ntaps > 0, "urse_filtfilt_fir: bad argument");
  const int P = ntaps - 1, padlen = 3 * ntaps;
  URSE_CHECK_ARG(lds_ >= ld + 2L * padlen + P, "urse_filtfilt_fir: scratch pitch %ld < %ld", (long)lds_, (long)(ld + 2L * padlen + P));
  hipStream_t st = (hipStream_t)stream;
  float* e = scratch;                       // [B, lds_] extended input of a pass
  float* f = scratch + (long)B * lds_;      // [B, lds_] its causal convolution
  const int gx = (int)((lds_ + 255) / 256 < 1024 ? (lds_ + 255) / 256 : 1024);
  const dim3 gfir((unsigned)((lds_ + FIR_OUT - 1) / FIR_OUT), B);
  hipLaunchKernelGGL(filtfilt_stage_kernel, dim3(gx, B), dim3(256), 0, st, x, lens, (long)ld, e, (long)lds_, P, padlen, 0);
  hipLaunchKernelGGL(fir_full_kernel, gfir, dim3(256), 0, st, (const float*)e, lens, (long)lds_, taps, ntaps_dev, (long)ntaps, 0, f,
                     P + 2 * padlen);
  hipLaunchKernelGGL(filtfilt_stage_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)f, lens, (long)lds_, e, (long)lds_, P, padlen, 1);
  hipLaunchKernelGGL(fir_full_kernel, gfir, dim3(256), 0, st, (const float*)e, lens, (long)lds_, taps, ntaps_dev, (long)ntaps, 0, f,
                     P + 2 * padlen);
  hipLaunchKernelGGL(filtfilt_crop_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)f, lens, (long)lds_, y, (long)ld, P, padlen);
  URSE_CHECK_LAUNCH("urse_filtfilt_fir");
  return URSE_OK;
}

extern "C" int urse_quantile_clip(float* x, const int32_t* lens, int B, int64_t ld, const float* qmin, const float* qmax,
                                  float* bounds, void* stream) {
  URSE_CHECK_ARG(x && lens && qmin && qmax && bounds && B > 0 && ld > 0, "urse_quantile_clip: bad argument");
  hipLaunchKernelGGL(quantile_clip_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, lens, (long)ld, qmin, qmax, bounds);
  URSE_CHECK_LAUNCH("urse_quantile_clip");
  return URSE_OK;
}

extern "C" int urse_zero_segments(float* x, int64_t ld, const int32_t* segments, int nseg, void* stream) {
  URSE_CHECK_ARG(x && ld > 0 && (segments || nseg == 0) && nseg >= 0, "urse_zero_segments: bad argument");
  if (nseg == 0) return URSE_OK;
  hipLaunchKernelGGL(zero_segments_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, x, (long)ld, segments, nseg);
  URSE_CHECK_LAUNCH("urse_zero_segments");
  return URSE_OK;
}

extern "C" int urse_joint_peak_scale(float* speech, float* noisy, float* noise, int B, int64_t ld, float target,
                                     void* peak_scratch, void* stream) {
  URSE_CHECK_ARG(speech && noisy && noise && peak_scratch && B > 0 && ld > 0, "urse_joint_peak_scale: bad argument");
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(peak_scratch, 0, sizeof(unsigned) * B, st);
  const int gx = (int)((ld + 255) / 256 < 512 ? (ld + 255) / 256 : 512);
  hipLaunchKernelGGL(absmax3_kernel, dim3(gx, B), dim3(256), 0, st, (const float*)speech, (const float*)noisy, (const float*)noise,
                     (long)ld, (unsigned*)peak_scratch);
  hipLaunchKernelGGL(scale3_kernel, dim3(gx, B), dim3(256), 0, st, speech, noisy, noise, (long)ld, (const unsigned*)peak_scratch,
                     target, 1e-6f);
  URSE_CHECK_LAUNCH("urse_joint_peak_scale");
  return URSE_OK;
}
