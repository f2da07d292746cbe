// Fourier resampling of whole utterances: y = scipy.signal.resample(x, num) for real f32 rows of ARBITRARY length.
//
// Reference call site: simulation/simulate_data_from_param.py:233-252 `bandwidth_limitation(..., res_type="scipy")`, i.e.
// librosa.resample(res_type="scipy") = scipy.signal.resample(y, ceil(n * ratio)): rfft of the whole utterance, the spectrum cut
// (or zero-padded) to num // 2 + 1 bins with the Nyquist bin doubled when it is cut and halved when it is introduced, irfft to
// `num` samples, scaled by num / n.  The two transform lengths are the utterance's own (192,000, 63,999, a prime ...), so they
// run as Bluestein (chirp-z) transforms: a length-n DFT is the chirp-modulated input convolved with the conjugate chirp, and the
// convolution is done with power-of-two FFTs of M >= 2n - 1 points.
//
// One 256-thread workgroup per row.  The M-point FFTs (M up to 2^20) are the four-step transforms of csrc/pesq_core.h
// (N1 x 1024: column transforms through LDS, twiddle, 1024-point row transforms through LDS; the forward one leaves its output
// in a permuted order, the inverse one takes it - a pointwise product does not care), ping-ponging between two M-point complex
// buffers of the caller's workspace.  The chirp w[k] = exp(-i pi k^2 / n) is evaluated from the EXACT phase k^2 mod 2n (64-bit
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

// plan[0 .. n): chirp w[k]; plan[n .. n + M): transform (permuted order) of b, b[k] = b[M - k] = conj(w[k]) for k < n, else 0
__global__ void __launch_bounds__(256) fa_plan_kernel(float2* __restrict__ plan, float2* __restrict__ tmp, int n, int M, const float2* tw) {
  using namespace pesq;
  __shared__ float2 s_la[1024], s_lb[1024];
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = nullptr; T.ired = nullptr;
  Params P;
  P.tw = tw; P.twn = FA_TWN;
  float2* w = plan;
  float2* bh = plan + n;
  for (int k = T.tid; k < M; k += T.nt) tmp[k] = make_float2(0.f, 0.f);
  T.sync();
  for (int k = T.tid; k < n; k += T.nt) {
    const long long r = ((long long)k * k) % (2LL * n);
    double s, c;
    sincospi((double)r / (double)n, &s, &c);
    w[k] = make_float2((float)c, (float)-s);
    const float2 cw = make_float2((float)c, (float)s);
    tmp[k] = cw;
    if (k) tmp[M - k] = cw;
  }
  T.sync();
  fft_big_forward(T, P, tmp, bh, M, s_la, s_lb);
}

struct FaArgs {
  const float* x; long ldx;       // [P, n] input rows
  float* y; long ldy;             // [P, num] output rows
  const float2* plan1;            // n-point plan (chirp + transformed conjugate chirp, M1)
  const float2* plan2;            // num-point plan (M2)
  char* ws; long ws_row;          // per row: 2 * max(M1, M2) float2 + (num / 2 + 1) float2
  int n, num, M1, M2;
  const float2* tw;
};

// Bluestein DFT of `len` points whose chirp-modulated input is already in bufA[0 .. M) (zero beyond len): result c in bufA (natural
// order, not yet multiplied by the chirp or divided by M)
__device__ void fa_convolve(const pesq::Team& T, const pesq::Params& P, float2* bufA, float2* bufB, const float2* bh, int M,
                            float2* la, float2* lb) {
  pesq::fft_big_forward(T, P, bufA, bufB, M, la, lb);
  for (int k = T.tid; k < M; k += T.nt) bufB[k] = fa_cmul(bufB[k], bh[k]);
  T.sync();
  pesq::fft_big_inverse(T, P, bufB, bufA, M, la, lb);
}

__global__ void __launch_bounds__(256) fa_resample_kernel(FaArgs a) {
  using namespace pesq;
  __shared__ float2 s_la[1024], s_lb[1024];
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = nullptr; T.ired = nullptr;
  Params P;
  P.tw = a.tw; P.twn = FA_TWN;
  const int row = blockIdx.x, n = a.n, num = a.num;
  const int Mx = a.M1 > a.M2 ? a.M1 : a.M2;
  float2* bufA = reinterpret_cast<float2*>(a.ws + (long)row * a.ws_row);
  float2* bufB = bufA + Mx;
  float2* Y = bufB + Mx;                                   // num / 2 + 1 bins of the resampled spectrum
  const float* x = a.x + (long)row * a.ldx;
  // ---- X = DFT_n(x), bins 0 .. N / 2 with N = min(n, num) ----
  const float2* w1 = a.plan1;
  for (int k = T.tid; k < a.M1; k += T.nt) bufA[k] = k < n ? make_float2(x[k] * w1[k].x, x[k] * w1[k].y) : make_float2(0.f, 0.f);
  T.sync();
  fa_convolve(T, P, bufA, bufB, a.plan1 + n, a.M1, s_la, s_lb);
  const int N = n < num ? n : num, nyq = N / 2 + 1, nb = num / 2 + 1;
  const float inv1 = 1.0f / (float)a.M1;
  for (int k = T.tid; k < nb; k += T.nt) {
    float2 v = make_float2(0.f, 0.f);
    if (k < nyq) {
      v = fa_cmul(w1[k], bufA[k]);
      v.x *= inv1; v.y *= inv1;
      if ((N & 1) == 0 && k == N / 2) {                    // the bin at the cut: both halves of the spectrum fold into it / it is split
        const float s = num < n ? 2.0f : (n < num ? 0.5f : 1.0f);
        v.x *= s; v.y *= s;
      }
    }
    Y[k] = v;
  }
  T.sync();
  // ---- y = irfft(Y, num) * num / n = Re(conj(DFT_num(conj(Z)))) / n, Z = the Hermitian extension of Y (irfft ignores the imaginary
  //      parts of bin 0 and, for even num, of bin num / 2) ----
  const float2* w2 = a.plan2;
  for (int k = T.tid; k < a.M2; k += T.nt) {
    float2 v = make_float2(0.f, 0.f);
    if (k < num) {
      const int kk = k <= num / 2 ? k : num - k;
      float2 z = Y[kk];
      if (k <= num / 2) z.y = -z.y;                        // conj(Z[k]); for k > num / 2, Z[k] = conj(Y[num - k]) and its conjugate is Y[num - k]
      if (kk == 0 || ((num & 1) == 0 && kk == num / 2)) z.y = 0.f;
      v = fa_cmul(z, w2[k]);
    }
    bufA[k] = v;
  }
  T.sync();
  fa_convolve(T, P, bufA, bufB, a.plan2 + num, a.M2, s_la, s_lb);
  const float sc = 1.0f / ((float)a.M2 * (float)n);
  float* y = a.y + (long)row * a.ldy;
  for (int m = T.tid; m < num; m += T.nt) {
    const float2 c = fa_cmul(w2[m], bufA[m]);
    y[m] = c.x * sc;
  }
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

extern "C" int urse_fft_resample_plan(void* plan, void* tmp, int n, void* stream) {
  URSE_CHECK_ARG(plan && tmp && n >= 2 && n <= (1 << 19), "urse_fft_resample_plan: bad argument");
  const float2* tw = fa_twiddles();
  if (!tw) { set_error("urse_fft_resample_plan: could not build the twiddle table"); return URSE_ERR_RUNTIME; }
  const int M = fa_pow2(2L * n - 1);
  hipLaunchKernelGGL(fa_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (float2*)plan, (float2*)tmp, n, M, tw);
  URSE_CHECK_LAUNCH("urse_fft_resample_plan");
  return URSE_OK;
}

extern "C" int urse_fft_resample_workspace_bytes(int P, int n, int num, int64_t* bytes) {
  URSE_CHECK_ARG(P > 0 && n >= 2 && num >= 2 && n <= (1 << 19) && num <= (1 << 19) && bytes, "urse_fft_resample_workspace_bytes: bad argument");
  const long M1 = fa_pow2(2L * n - 1), M2 = fa_pow2(2L * num - 1), Mx = M1 > M2 ? M1 : M2;
  const long row = ((2 * Mx + num / 2 + 1) * 8 + 63) / 64 * 64;
  *bytes = row * P;
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
  FaArgs a;
  a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.plan1 = (const float2*)plan_n; a.plan2 = (const float2*)plan_num;
  a.ws = (char*)workspace; a.ws_row = need / P; a.n = n; a.num = num; a.M1 = fa_pow2(2L * n - 1); a.M2 = fa_pow2(2L * num - 1); a.tw = tw;
  hipLaunchKernelGGL(fa_resample_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, a);
  URSE_CHECK_LAUNCH("urse_fft_resample");
  return URSE_OK;
}
