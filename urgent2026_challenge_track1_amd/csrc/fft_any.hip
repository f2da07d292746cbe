// Fourier resampling of whole utterances: y = scipy.signal.resample(x, num) for real f32 rows of ARBITRARY length.
//
// Reference call site: simulation/simulate_data_from_param.py:233-252 `bandwidth_limitation(..., res_type="scipy")`, i.e.
// librosa.resample(res_type="scipy") = scipy.signal.resample(y, ceil(n * ratio)): rfft of the whole utterance, the spectrum cut
// (or zero-padded) to num // 2 + 1 bins with the Nyquist bin doubled when it is cut and halved when it is introduced, irfft to
// `num` samples, scaled by num / n.  The two transform lengths are the utterance's own (192,000, 63,999, a prime ...), so they
// run as Bluestein (chirp-z) transforms: a length-n DFT is the chirp-modulated input convolved with the conjugate chirp, and the
// convolution is done with power-of-two FFTs of M >= 2n - 1 points.
//
// The M-point FFTs (M up to 2^20) are the four-step transforms of csrc/pesq_core.h (N1 x 1024: column transforms through LDS,
// twiddle, 1024-point row transforms through LDS; the forward one leaves its output in a permuted order, the inverse one takes
// it - a pointwise product does not care), here with one LAUNCH per step and a workgroup per column tile / per row, ping-ponging
// between two M-point complex buffers per utterance of the caller's workspace.  The chirp w[k] = exp(-i pi k^2 / n) is evaluated from the EXACT phase k^2 mod 2n (64-bit
// integers, float64 sincos) and stored as float2; its transform is built once per (n, M) by urse_fft_resample_plan and cached by the
// host.  Everything after that is float32 like scipy's own pocketfft path on float32 input.
#include <math.h>

#include <mutex>
#include <vector>

#include "urse_common.h"
#include "pesq_core.h"

namespace urse {

constexpr int FA_TWN = 1 << 20;          // twiddle table: exp(-2 pi i k / 2^20), k < 2^19 (4 MB per device, built on first use)

static std::mutex g_fa_mutex;
static float2* g_fa_tw[16] = {nullptr};

static const float2* fa_twiddles() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_fa_mutex);
  if (!g_fa_tw[dev]) {
    std::vector<float2> h(FA_TWN / 2);
    for (int k = 0; k < FA_TWN / 2; ++k) {
      const double ang = -2.0 * M_PI * k / FA_TWN;
      h[k].x = (float)cos(ang); h[k].y = (float)sin(ang);
    }
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_fa_tw[dev] = d;
  }
  return g_fa_tw[dev];
}

__device__ __forceinline__ float2 fa_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// ---- the M = N1 x 1024 four-step FFT of pesq_core.h, one step per launch, a workgroup per column tile / per row ---------------
// (one workgroup per utterance - the first version - kept a CU for ~60 ms per call: the train step's cooperative forward kernel
//  then waited for that CU, and the simulator no longer fitted beside the step: 253 vs 172 ms per step with dynamic mixing)
// Buffers are [rows of the batch][M] float2; blockIdx.y = row of the batch.
struct FaFft {
  float2* a;            // natural-order input of the forward transform / output of the inverse one
  float2* b;            // permuted-order spectrum (pos(k) = (k mod N1) * 1024 + k / N1)
  const float2* bh;     // transformed conjugate chirp / filter (permuted order), or null
  long bh_stride;       // elements between the filters of consecutive rows of the batch (0: one filter for all)
  const float2* tw;
  long stride;          // elements between consecutive rows of the batch (>= M)
  int M;
};

// columns, forward: N1-point transforms down `cols` columns at a time, twiddle, into b
__global__ void __launch_bounds__(256) fa_cols_fwd_kernel(FaFft q) {
  using namespace pesq;
  __shared__ float2 la[1024], lb[1024];
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = nullptr; T.ired = nullptr;
  Params P;
  P.tw = q.tw; P.twn = FA_TWN;
  const int n = q.M, n1 = n >> 10, cols = 1024 / n1, tstep = FA_TWN / n, c0 = blockIdx.x * cols;
  const float2* src = q.a + (long)blockIdx.y * q.stride;
  float2* dst = q.b + (long)blockIdx.y * q.stride;
  for (int i = T.tid; i < 1024; i += T.nt) {
    const int r = i / cols, c = i - r * cols;
    la[c * n1 + r] = src[1024 * r + c0 + c];
  }
  T.sync();
  float2* R = fft_batched(T, la, lb, n1, cols, false, P);
  for (int i = T.tid; i < 1024; i += T.nt) {
    const int k1 = i / cols, c = i - k1 * cols, n2 = c0 + c;
    const float2 v = R[c * n1 + k1];
    const int e = n2 * k1;
    float2 w = P.tw[(e < n / 2 ? e : e - n / 2) * tstep];
    if (e >= n / 2) { w.x = -w.x; w.y = -w.y; }
    dst[k1 * 1024 + n2] = fa_cmul(v, w);
  }
}

// rows: 1024-point forward transform of row k1 of b in place; with bh: x bh, 1024-point INVERSE transform, x conj twiddle - the first
// half of the inverse four-step transform, which works on the same rows - so that a convolution crosses memory three times, not five
__global__ void __launch_bounds__(256) fa_rows_kernel(FaFft q) {
  using namespace pesq;
  __shared__ float2 la[1024], lb[1024];
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = nullptr; T.ired = nullptr;
  Params P;
  P.tw = q.tw; P.twn = FA_TWN;
  const int n = q.M, k1 = blockIdx.x, tstep = FA_TWN / n;
  float2* row = q.b + (long)blockIdx.y * q.stride + (long)k1 * 1024;
  for (int i = T.tid; i < 1024; i += T.nt) la[i] = row[i];
  T.sync();
  float2* R = fft(T, la, lb, 1024, false, P);
  if (q.bh == nullptr) {
    for (int i = T.tid; i < 1024; i += T.nt) row[i] = R[i];
    return;
  }
  float2* O = R == la ? lb : la;
  const float2* bh = q.bh + (long)blockIdx.y * q.bh_stride + (long)k1 * 1024;
  for (int i = T.tid; i < 1024; i += T.nt) O[i] = fa_cmul(R[i], bh[i]);
  T.sync();
  float2* S = fft(T, O, R, 1024, true, P);
  for (int n2 = T.tid; n2 < 1024; n2 += T.nt) {
    const float2 v = S[n2];
    const int e = n2 * k1;
    float2 w = P.tw[(e < n / 2 ? e : e - n / 2) * tstep];
    if (e >= n / 2) { w.x = -w.x; w.y = -w.y; }
    row[n2] = make_float2(v.x * w.x + v.y * w.y, v.y * w.x - v.x * w.y);      // times conj(w)
  }
}

// columns, inverse: N1-point inverse transforms down the columns of b into a (natural order, not scaled)
__global__ void __launch_bounds__(256) fa_cols_inv_kernel(FaFft q) {
  using namespace pesq;
  __shared__ float2 la[1024], lb[1024];
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = nullptr; T.ired = nullptr;
  Params P;
  P.tw = q.tw; P.twn = FA_TWN;
  const int n = q.M, n1 = n >> 10, cols = 1024 / n1, c0 = blockIdx.x * cols;
  const float2* src = q.b + (long)blockIdx.y * q.stride;
  float2* dst = q.a + (long)blockIdx.y * q.stride;
  for (int i = T.tid; i < 1024; i += T.nt) {
    const int k1 = i / cols, c = i - k1 * cols;
    la[c * n1 + k1] = src[k1 * 1024 + c0 + c];
  }
  T.sync();
  float2* R = fft_batched(T, la, lb, n1, cols, true, P);
  for (int i = T.tid; i < 1024; i += T.nt) {
    const int r = i / cols, c = i - r * cols;
    dst[1024 * r + c0 + c] = R[c * n1 + r];
  }
}

// plan, stage 1: chirp w[k] = exp(-i pi k^2 / n) (exact phase k^2 mod 2n, float64 sincos) and the sequence b[k] = b[M - k] = conj(w[k])
__global__ void __launch_bounds__(256) fa_chirp_kernel(float2* __restrict__ w, float2* __restrict__ tmp, int n, int M) {
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    const int kk = k < n ? k : (M - k < n ? M - k : -1);
    if (kk >= 0) {
      const long long r = ((long long)kk * kk) % (2LL * n);
      double s, c;
      sincospi((double)r / (double)n, &s, &c);
      v = make_float2((float)c, (float)s);
      if (k < n) w[k] = make_float2((float)c, (float)-s);
    }
    tmp[k] = v;
  }
}

struct FaArgs {
  const float* x; long ldx;       // [P, n] input rows
  float* y; long ldy;             // [P, num] output rows
  const float2* plan1;            // n-point plan (chirp + transformed conjugate chirp, M1)
  const float2* plan2;            // num-point plan (M2)
  float2* bufA; float2* bufB;     // [P][Mx] each
  float2* Y;                      // [P][num / 2 + 1] resampled spectrum
  long stride;                    // Mx
  int n, num, M1, M2;
};

// a[k] = x[k] w1[k] (k < n), zero up to M1
__global__ void __launch_bounds__(256) fa_prep1_kernel(FaArgs a) {
  const int row = blockIdx.y;
  const float* x = a.x + (long)row * a.ldx;
  float2* A = a.bufA + (long)row * a.stride;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.M1; k += gridDim.x * blockDim.x)
    A[k] = k < a.n ? make_float2(x[k] * a.plan1[k].x, x[k] * a.plan1[k].y) : make_float2(0.f, 0.f);
}

// X = w1 c / M1 on bins 0 .. N / 2 (N = min(n, num)) -> Y with the rule of the bin at the cut; then the chirp-modulated conjugate of
// Y's Hermitian extension -> bufA (zero up to M2), the input of the second transform
__global__ void __launch_bounds__(256) fa_mid_kernel(FaArgs a) {
  const int row = blockIdx.y, n = a.n, num = a.num;
  const float2* C = a.bufA + (long)row * a.stride;
  float2* Y = a.Y + (long)row * (num / 2 + 1);
  const int N = n < num ? n : num, nyq = N / 2 + 1, nb = num / 2 + 1;
  const float inv1 = 1.0f / (float)a.M1;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nb; k += gridDim.x * blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    if (k < nyq) {
      v = fa_cmul(a.plan1[k], C[k]);
      v.x *= inv1; v.y *= inv1;
      if ((N & 1) == 0 && k == N / 2) {                    // the bin at the cut: both halves of the spectrum fold into it / it is split
        const float s = num < n ? 2.0f : (n < num ? 0.5f : 1.0f);
        v.x *= s; v.y *= s;
      }
    }
    Y[k] = v;
  }
}

__global__ void __launch_bounds__(256) fa_prep2_kernel(FaArgs a) {
  const int row = blockIdx.y, num = a.num;
  const float2* Y = a.Y + (long)row * (num / 2 + 1);
  float2* A = a.bufA + (long)row * a.stride;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < a.M2; k += gridDim.x * blockDim.x) {
    float2 v = make_float2(0.f, 0.f);
    if (k < num) {
      const int kk = k <= num / 2 ? k : num - k;
      float2 z = Y[kk];
      if (k <= num / 2) z.y = -z.y;                        // conj(Z[k]); for k > num / 2, Z[k] = conj(Y[num - k]) and its conjugate is Y[num - k]
      if (kk == 0 || ((num & 1) == 0 && kk == num / 2)) z.y = 0.f;     // irfft ignores these imaginary parts
      v = fa_cmul(z, a.plan2[k]);
    }
    A[k] = v;
  }
}

// y = irfft(Y, num) * num / n = Re(w2 c) / (M2 n)
__global__ void __launch_bounds__(256) fa_post_kernel(FaArgs a) {
  const int row = blockIdx.y;
  const float2* C = a.bufA + (long)row * a.stride;
  float* y = a.y + (long)row * a.ldy;
  const float sc = 1.0f / ((float)a.M2 * (float)a.n);
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < a.num; m += gridDim.x * blockDim.x) {
    const float2 c = fa_cmul(a.plan2[m], C[m]);
    y[m] = c.x * sc;
  }
}

// ---- linear convolution of whole utterances with (per-utterance) filters by the same transforms (add_reverberation) -----------------
// a[b][k] = x[b][k] for k < len[b], 0 up to M
__global__ void __launch_bounds__(256) fa_load_real_kernel(const float* __restrict__ x, const int* __restrict__ lens, long ld, int fixed_len,
                                                           float2* __restrict__ a, long stride, int M) {
  const int b = blockIdx.y, n = lens ? lens[b] : fixed_len;
  const float* xr = x + (long)b * ld;
  float2* ar = a + (long)b * stride;
  for (int k = blockIdx.x * 1024 + threadIdx.x; k < min(M, (int)(blockIdx.x + 1) * 1024); k += 256) ar[k] = make_float2(k < n ? xr[k] : 0.f, 0.f);
}
// y[b][k] = Re(a[b][k]) / M for k < len[b], 0 up to `width`
__global__ void __launch_bounds__(256) fa_store_real_kernel(const float2* __restrict__ a, long stride, const int* __restrict__ lens, float* __restrict__ y,
                                                            long ld, int width, int M) {
  const int b = blockIdx.y, n = lens[b];
  const float2* ar = a + (long)b * stride;
  float* yr = y + (long)b * ld;
  const float sc = 1.0f / (float)M;
  for (int k = blockIdx.x * 1024 + threadIdx.x; k < min(width, (int)(blockIdx.x + 1) * 1024); k += 256) yr[k] = k < n ? ar[k].x * sc : 0.f;
}

static int fa_pow2(long v) {
  long m = 2048;                                           // (the four-step transform needs N1 >= 2)
  while (m < v) m <<= 1;
  return (int)m;
}

}  // namespace urse

using namespace urse;

extern "C" int urse_fft_resample_plan_elems(int n, int64_t* elems, int64_t* tmp_elems) {
  URSE_CHECK_ARG(n >= 2 && n <= (1 << 19) && elems && tmp_elems, "urse_fft_resample_plan_elems: length %d out of range (2 .. 2^19)", n);
  const int M = fa_pow2(2L * n - 1);
  *elems = (int64_t)n + M;
  *tmp_elems = M;
  return URSE_OK;
}

static void fa_convolve(FaFft q, int P, hipStream_t st) {
  const int n1 = q.M >> 10, cols = 1024 / n1;
  hipLaunchKernelGGL(fa_cols_fwd_kernel, dim3(1024 / cols, P), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fa_rows_kernel, dim3(n1, P), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fa_cols_inv_kernel, dim3(1024 / cols, P), dim3(256), 0, st, q);
}

extern "C" int urse_fft_resample_plan(void* plan, void* tmp, int n, void* stream) {
  URSE_CHECK_ARG(plan && tmp && n >= 2 && n <= (1 << 19), "urse_fft_resample_plan: bad argument");
  const float2* tw = fa_twiddles();
  if (!tw) { set_error("urse_fft_resample_plan: could not build the twiddle table"); return URSE_ERR_RUNTIME; }
  const int M = fa_pow2(2L * n - 1);
  hipStream_t st = (hipStream_t)stream;
  float2* w = (float2*)plan;
  hipLaunchKernelGGL(fa_chirp_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, w, (float2*)tmp, n, M);
  FaFft q;
  q.a = (float2*)tmp; q.b = w + n; q.bh = nullptr; q.bh_stride = 0; q.tw = tw; q.stride = M; q.M = M;
  const int n1 = M >> 10, cols = 1024 / n1;
  hipLaunchKernelGGL(fa_cols_fwd_kernel, dim3(1024 / cols, 1), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fa_rows_kernel, dim3(n1, 1), dim3(256), 0, st, q);
  URSE_CHECK_LAUNCH("urse_fft_resample_plan");
  return URSE_OK;
}

extern "C" int urse_fft_resample_workspace_bytes(int P, int n, int num, int64_t* bytes) {
  URSE_CHECK_ARG(P > 0 && n >= 2 && num >= 2 && n <= (1 << 19) && num <= (1 << 19) && bytes, "urse_fft_resample_workspace_bytes: bad argument");
  const long M1 = fa_pow2(2L * n - 1), M2 = fa_pow2(2L * num - 1), Mx = M1 > M2 ? M1 : M2;
  *bytes = ((long)P * (2 * Mx + num / 2 + 1) * 8 + 63) / 64 * 64;
  return URSE_OK;
}

extern "C" int urse_fft_resample(const float* x, int64_t ldx, float* y, int64_t ldy, const void* plan_n, const void* plan_num,
                                 void* workspace, int64_t workspace_bytes, int P, int n, int num, void* stream) {
  URSE_CHECK_ARG(x && y && plan_n && plan_num && workspace && P > 0 && n >= 2 && num >= 2 && n <= (1 << 19) && num <= (1 << 19) &&
                     ldx >= n && ldy >= num,
                 "urse_fft_resample: bad argument");
  int64_t need = 0;
  int rc = urse_fft_resample_workspace_bytes(P, n, num, &need);
  if (rc) return rc;
  URSE_CHECK_ARG(workspace_bytes >= need, "urse_fft_resample: workspace of %ld bytes, %ld needed", (long)workspace_bytes, (long)need);
  const float2* tw = fa_twiddles();
  if (!tw) { set_error("urse_fft_resample: could not build the twiddle table"); return URSE_ERR_RUNTIME; }
  hipStream_t st = (hipStream_t)stream;
  FaArgs a;
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.plan1 = (const float2*)plan_n; a.plan2 = (const float2*)plan_num;
  a.n = n; a.num = num; a.M1 = fa_pow2(2L * n - 1); a.M2 = fa_pow2(2L * num - 1);
  const long Mx = a.M1 > a.M2 ? a.M1 : a.M2;
  a.stride = Mx;
  a.bufA = (float2*)workspace; a.bufB = a.bufA + (long)P * Mx; a.Y = a.bufB + (long)P * Mx;
  FaFft q;
  q.a = a.bufA; q.b = a.bufB; q.tw = tw; q.stride = Mx; q.bh_stride = 0;
  hipLaunchKernelGGL(fa_prep1_kernel, dim3(ceil_div(a.M1, 1024), P), dim3(256), 0, st, a);
  q.bh = a.plan1 + n; q.M = a.M1;
  fa_convolve(q, P, st);
  hipLaunchKernelGGL(fa_mid_kernel, dim3(ceil_div(num / 2 + 1, 1024), P), dim3(256), 0, st, a);
  hipLaunchKernelGGL(fa_prep2_kernel, dim3(ceil_div(a.M2, 1024), P), dim3(256), 0, st, a);
  q.bh = a.plan2 + num; q.M = a.M2;
  fa_convolve(q, P, st);
  hipLaunchKernelGGL(fa_post_kernel, dim3(ceil_div(num, 1024), P), dim3(256), 0, st, a);
  URSE_CHECK_LAUNCH("urse_fft_resample");
  return URSE_OK;
}

// y[b, :len_b] = scipy.signal.convolve(x[b, :len_b], taps[b or 0, :ntaps], "full")[:len_b] by power-of-two FFTs of M >= max_len +
// max_ntaps - 1 points (add_reverberation, simulate_data_from_param.py:220-230: the direct form costs len x ntaps, 1.3 ms per call
// at 4 s x 1 s @ 48 kHz).  Workspace: three [B][M] complex buffers.
extern "C" int urse_fft_convolve_workspace_bytes(int B, int max_len, int max_ntaps, int64_t* bytes) {
  URSE_CHECK_ARG(B > 0 && max_len > 0 && max_ntaps > 0 && (long)max_len + max_ntaps - 1 <= (1 << 20) && bytes,
                 "urse_fft_convolve_workspace_bytes: bad argument (len + ntaps - 1 <= 2^20)");
  const long M = fa_pow2((long)max_len + max_ntaps - 1);
  *bytes = (long)B * 3 * M * 8;
  return URSE_OK;
}

extern "C" int urse_fft_convolve(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps, int64_t ldt,
                                 int taps_per_utt, float* y, int max_len, int max_ntaps, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  URSE_CHECK_ARG(x && lens && taps && ntaps && y && workspace && B > 0 && x != y && ld >= max_len && ldt >= max_ntaps,
                 "urse_fft_convolve: bad argument");
  int64_t need = 0;
  int rc = urse_fft_convolve_workspace_bytes(B, max_len, max_ntaps, &need);
  if (rc) return rc;
  URSE_CHECK_ARG(workspace_bytes >= need, "urse_fft_convolve: workspace of %ld bytes, %ld needed", (long)workspace_bytes, (long)need);
  const float2* tw = fa_twiddles();
  if (!tw) { set_error("urse_fft_convolve: could not build the twiddle table"); return URSE_ERR_RUNTIME; }
  hipStream_t st = (hipStream_t)stream;
  const int M = fa_pow2((long)max_len + max_ntaps - 1);
  float2* bufA = (float2*)workspace;
  float2* bufB = bufA + (long)B * M;
  float2* Hh = bufB + (long)B * M;
  const int n1 = M >> 10, cols = 1024 / n1, PF = taps_per_utt ? B : 1;
  FaFft q;
  q.tw = tw; q.stride = M; q.M = M;
  // filters -> permuted-order spectra
  hipLaunchKernelGGL(fa_load_real_kernel, dim3(M / 1024, PF), dim3(256), 0, st, taps, ntaps, (long)ldt, 0, bufA, (long)M, M);
  q.a = bufA; q.b = Hh; q.bh = nullptr; q.bh_stride = 0;
  hipLaunchKernelGGL(fa_cols_fwd_kernel, dim3(1024 / cols, PF), dim3(256), 0, st, q);
  hipLaunchKernelGGL(fa_rows_kernel, dim3(n1, PF), dim3(256), 0, st, q);
  // signals: forward transform, x filter, inverse transform
  hipLaunchKernelGGL(fa_load_real_kernel, dim3(M / 1024, B), dim3(256), 0, st, x, lens, (long)ld, 0, bufA, (long)M, M);
  q.a = bufA; q.b = bufB; q.bh = Hh; q.bh_stride = taps_per_utt ? M : 0;
  fa_convolve(q, B, st);
  hipLaunchKernelGGL(fa_store_real_kernel, dim3(ceil_div(max_len, 1024), B), dim3(256), 0, st, bufA, (long)M, lens, y, (long)ld, max_len, M);
  URSE_CHECK_LAUNCH("urse_fft_convolve");
  return URSE_OK;
}
