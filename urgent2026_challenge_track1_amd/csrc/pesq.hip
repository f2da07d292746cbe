// Batched PESQ on the GPU: one 256-thread workgroup per (reference, degraded) pair runs the whole ITU-T P.862 measurement
// (pesq_core.h) on a per-pair slice of the caller's workspace.  Replaces pesq.pesq(fs, ref, deg, mode,
// on_error=RETURN_VALUES) behind evaluation_metrics/calculate_intrusive_se_metrics.py:52-88 (a process pool of one CPU core
// per pair there).  Pairs are independent: the grid is the batch, 2-4 workgroups share a CU (36 KB of LDS each), the long
// FFTs ping-pong in the pair's own 2 MB of workspace, which stays in L2 / Infinity Cache while the pair is being measured.
#include <math.h>

#include <mutex>
#include <vector>

#include "urse_common.h"
#include "pesq_core.h"

namespace urse {

struct PesqArgs {
  const float* ref; const float* deg; long ld;
  const int32_t* lens;
  int L, fs, wb;
  float* mos; float* raw; int32_t* trace;
  char* ws; long ws_pair;            // bytes per pair
  const float2* tw; int twn;
  int na, nw, nfr, p2;
};

__host__ __device__ inline long pesq_align(long v) { return (v + 63) / 64 * 64; }

#ifndef URSE_PESQ_WAVES_PER_SIMD
#define URSE_PESQ_WAVES_PER_SIMD 4      // workgroups per CU the register budget is cut for (one wave of each per SIMD): 120 VGPRs and
                                        // 37 KB of LDS let four pairs share a CU - 13.1 k pairs/s at 2,048 pairs per launch against 12.0 k
                                        // with three (132 VGPRs); launches of <= 768 pairs are 7 % slower
#endif
__global__ void __launch_bounds__(256, URSE_PESQ_WAVES_PER_SIMD) pesq_kernel(PesqArgs a) {
  using namespace pesq;
  __shared__ float2 s_la[1024], s_lb[1024];
  __shared__ float s_x[1024], s_h[1024], s_w[2048], s_iir[512];
  __shared__ double s_red[256];
  __shared__ int s_ired[256];
  const int pair = blockIdx.x;
  Team T;
  T.tid = threadIdx.x; T.nt = blockDim.x; T.red = s_red; T.ired = s_ired;
  Lds Ld;
  Ld.la = s_la; Ld.lb = s_lb; Ld.x = s_x; Ld.h = s_h; Ld.w = s_w; Ld.wcap = 2048; Ld.iir = s_iir;
  Params P;
  P.fs = a.fs; P.wb = a.wb; P.ds = a.fs == 8000 ? 32 : 64; P.align_nfft = a.fs == 8000 ? 512 : 1024;
  P.pad = DATAPADDING_MSECS * (a.fs / 1000);
  P.tb = a.fs == 8000 ? &TABLES_8K : &TABLES_16K;
  P.nb = P.tb->nb; P.tw = a.tw; P.twn = a.twn;
  const int sb = SEARCHBUFFER * P.ds;
  int len = a.lens ? a.lens[pair] : a.L;
  if (len > a.L) len = a.L;
  Pair S;
  char* base = a.ws + (long)pair * a.ws_pair;
  float* p = reinterpret_cast<float*>(base);
  for (int s = 0; s < 2; ++s) {
    S.data[s] = p; p += a.na; S.adata[s] = p; p += a.na; S.vad[s] = p; p += a.nw; S.logvad[s] = p; p += a.nw;
    S.nsamp[s] = len + 2 * sb;
  }
  S.na = a.na;
  S.tweaked = p; p += a.na; S.doubly = p; p += a.na;
  S.ppd_ref = p; p += (long)a.nfr * 49; S.ppd_deg = p; p += (long)a.nfr * 49;
  S.fd = p; p += a.nfr; S.fda = p; p += a.nfr; S.tpr = p; p += a.nfr;
  S.scratch = p; p += 8L * a.nfr + 4096;
  S.fst = p; p += F_COUNT;
  S.st = reinterpret_cast<int*>(p); p += I_COUNT;
  S.ca = reinterpret_cast<float2*>(base + pesq_align((char*)p - base));
  S.cb = S.ca + a.p2;
  S.p2max = a.p2;
  // SIGNAL_INFO.data: search buffer of zeros, the samples on the 16-bit scale (both signals divided by their common peak when
  // it exceeds 1, as the package's wrapper does), zeros
  const float* r = a.ref + (long)pair * a.ld;
  const float* d = a.deg + (long)pair * a.ld;
  float mx = 0.f;
  for (int i = T.tid; i < len; i += T.nt) mx = fmaxf(mx, fmaxf(fabsf(r[i]), fabsf(d[i])));
  mx = T.maxf(mx);
  const float sc = 32768.f / (mx > 1.f ? mx : 1.f);
  for (int i = T.tid; i < a.na; i += T.nt) {
    const int j = i - sb;
    const bool in = j >= 0 && j < len;
    S.data[0][i] = in ? r[j] * sc : 0.f;
    S.data[1][i] = in ? d[j] * sc : 0.f;
  }
  T.sync();
  int32_t* trace = a.trace + (long)pair * TRACE_INTS;
  const float minlen = (float)(a.fs / 4);
  float raw, mos;
  if ((float)len < minlen) { raw = -1000.f; }
  else raw = pesq_pair(T, P, S, Ld, trace);
  if (raw <= -999.f) mos = __int_as_float(0x7fc00000);     // NO_UTTERANCES_DETECTED (or shorter than 1/4 s): NaN
  else mos = a.wb ? 0.999f + 4.0f / (1.0f + expf(-1.3669f * raw + 3.8224f)) : 0.999f + 4.0f / (1.0f + expf(-1.4945f * raw + 4.6607f));
  if (T.tid == 0) { a.mos[pair] = mos; if (a.raw) a.raw[pair] = raw; }
}

struct PesqPlan { int na, nw, nfr, p2; long ws_pair; };

static PesqPlan pesq_plan(int L, int fs) {
  const int ds = fs == 8000 ? 32 : 64, nfft = fs == 8000 ? 512 : 1024, pad = 320 * (fs / 1000), sb = 75 * ds;
  PesqPlan q;
  const int nsamp = L + 2 * sb;
  q.na = (nsamp + pad + 4 * nfft + 64 + 63) / 64 * 64;
  q.nw = (q.na / ds + 8 + 63) / 64 * 64;
  q.nfr = (q.na / (4 * ds) + 8 + 63) / 64 * 64;
  int p2 = 65536;                                  // (at least: see below)
  while (p2 < nsamp - 2 * sb + pad) p2 <<= 1;                    // the bad-interval realignment transforms up to 2 x (interval + 8 frames)
  q.p2 = p2;
  long floats = 2L * (2L * q.na + 2L * q.nw) + 2L * q.na + 2L * q.nfr * 49 + 3L * q.nfr + 8L * q.nfr + 4096 + pesq::F_COUNT +
                pesq::I_COUNT;
  q.ws_pair = pesq_align(floats * 4) + 2L * p2 * 8;
  q.ws_pair = pesq_align(q.ws_pair);
  return q;
}

static std::mutex g_tw_mutex;
static float2* g_tw[16] = {nullptr};
constexpr int PESQ_TWN = 1 << 18;

static const float2* pesq_twiddles() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_tw_mutex);
  if (!g_tw[dev]) {
    std::vector<float2> h(PESQ_TWN / 2);
    for (int k = 0; k < PESQ_TWN / 2; ++k) {
      const double ang = -2.0 * M_PI * k / PESQ_TWN;
      h[k].x = (float)cos(ang); h[k].y = (float)sin(ang);
    }
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    g_tw[dev] = d;
  }
  return g_tw[dev];
}

}  // namespace urse

using namespace urse;

extern "C" int urse_pesq_workspace_bytes(int pairs, int L, int fs, int64_t* bytes) {
  URSE_CHECK_ARG(bytes && pairs > 0 && L > 0 && (fs == 8000 || fs == 16000), "urse_pesq_workspace_bytes: bad argument");
  *bytes = pesq_plan(L, fs).ws_pair * pairs;
  return URSE_OK;
}

extern "C" int urse_pesq_batch(const float* ref, const float* deg, int64_t ld, const int32_t* lens, int pairs, int L, int fs, int wb,
                               float* mos, float* raw, int32_t* trace, void* workspace, int64_t workspace_bytes, void* stream) {
  URSE_CHECK_ARG(ref && deg && mos && trace && workspace && pairs > 0 && L > 0 && ld >= L, "urse_pesq_batch: bad argument");
  URSE_CHECK_ARG(fs == 8000 || fs == 16000, "urse_pesq_batch: fs must be 8000 or 16000 (got %d): resample first", fs);
  URSE_CHECK_ARG(!(wb && fs != 16000), "urse_pesq_batch: wide-band mode needs fs = 16000");
  URSE_CHECK_ARG(L < (1 << 22), "urse_pesq_batch: signals longer than 2^22 samples are not supported");
  const PesqPlan q = pesq_plan(L, fs);
  URSE_CHECK_ARG(q.p2 <= PESQ_TWN, "urse_pesq_batch: %d samples exceed the transform table", L);
  URSE_CHECK_ARG(workspace_bytes >= q.ws_pair * pairs, "urse_pesq_batch: workspace too small (%ld < %ld)", (long)workspace_bytes,
                 (long)(q.ws_pair * pairs));
  const float2* tw = pesq_twiddles();
  if (!tw) { set_error("urse_pesq_batch: could not build the twiddle table"); return URSE_ERR_RUNTIME; }
  PesqArgs a;
  a.ref = ref; a.deg = deg; a.ld = ld; a.lens = lens; a.L = L; a.fs = fs; a.wb = wb; a.mos = mos; a.raw = raw; a.trace = trace;
  a.ws = (char*)workspace; a.ws_pair = q.ws_pair; a.tw = tw; a.twn = PESQ_TWN; a.na = q.na; a.nw = q.nw; a.nfr = q.nfr; a.p2 = q.p2;
  hipLaunchKernelGGL(pesq_kernel, dim3(pairs), dim3(256), 0, (hipStream_t)stream, a);
  URSE_CHECK_LAUNCH("urse_pesq_batch");
  return URSE_OK;
}
