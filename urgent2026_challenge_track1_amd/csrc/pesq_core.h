// PESQ (ITU-T P.862 + P.862.1 / P.862.2 mappings) for one (reference, degraded) pair, written for a TEAM of threads:
// one 256-thread workgroup per pair on the GPU (pesq.hip).  Replaces pesq.pesq() behind
// evaluation_metrics/calculate_intrusive_se_metrics.py:52-88.
//
// The algorithm has three kinds of work and each gets the mapping that suits it:
//   * whole-signal DSP (level alignment and IRS filters = 2^17-point FFTs, VAD energies, envelope cross-correlations,
//     per-frame spectra and Bark densities): data-parallel loops over the team, block reductions, Stockham FFTs that
//     run in LDS for the 512 / 1024-point frames and in the (L2-resident) workspace for the long ones;
//   * recursive filters (13 second-order sections over 80 k samples): one wave, section s on lane s, samples handed from
//     lane to lane with a shuffle - a software pipeline over the cascade;
//   * integer bookkeeping (utterance windows, delay histogram peaks, utterance splitting, bad intervals): thread 0, with
//     the results broadcast through the per-pair state in memory.
// The same source also compiles for the host with a one-thread team (PQ_HOST): that build exists only so the control flow
// can be debugged without a GPU (scripts/pesq_host_debug.cpp); nothing in the package calls it.
#pragma once
#include <math.h>
#include <stdint.h>

#include "pesq_tables.h"

#if defined(__HIPCC__) && !defined(PQ_HOST)
#define PQ_FN __device__
#define PQ_DEVICE 1
#else
#define PQ_FN
#define PQ_DEVICE 0
struct float2 { float x, y; };
#endif

namespace pesq {

constexpr int SEARCHBUFFER = 75, DATAPADDING_MSECS = 320, MAXNUTT = 50, MINSPEECHLGTH = 4, JOINSPEECHLGTH = 50,
              MINUTTLENGTH = 50, MAXBAD = 64, TRACE_INTS = 8 + 3 * MAXNUTT + 2 * MAXBAD;
constexpr float TWOPI_F = 6.283185307179586f;

struct Team {
  int tid, nt;
  double* red;          // nt doubles of scratch shared by the team
  int* ired;            // nt ints
  PQ_FN void sync() const {
#if PQ_DEVICE
    __syncthreads();
#endif
  }
  PQ_FN double sum(double v) const {
#if PQ_DEVICE
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    sync();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    sync();
    double r = 0;
    for (int w = 0; w < (nt >> 6); ++w) r += red[w];
    sync();
    return r;
#else
    return v;
#endif
  }
  PQ_FN float maxf(float v) const {
#if PQ_DEVICE
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    sync();
    if ((tid & 63) == 0) red[tid >> 6] = (double)v;
    sync();
    float r = (float)red[0];
    for (int w = 1; w < (nt >> 6); ++w) r = fmaxf(r, (float)red[w]);
    sync();
    return r;
#else
    return v;
#endif
  }
  // first (lowest index) maximum of the candidates; idx < 0 = no candidate
  PQ_FN void argmax(float v, int idx, float* best, int* bidx) const {
#if PQ_DEVICE
    sync();
    red[tid] = (double)v;
    ired[tid] = idx;
    sync();
    for (int s = nt >> 1; s > 0; s >>= 1) {
      if (tid < s) {
        const double o = red[tid + s];
        const int oi = ired[tid + s];
        if (oi >= 0 && (ired[tid] < 0 || o > red[tid] || (o == red[tid] && oi < ired[tid]))) { red[tid] = o; ired[tid] = oi; }
      }
      sync();
    }
    *best = (float)red[0];
    *bidx = ired[0];
    sync();
#else
    *best = v;
    *bidx = idx;
#endif
  }
};

struct Params {
  int fs, wb;            // 8000 | 16000; wide-band mode
  int ds, align_nfft, pad, nb;
  const Tables* tb;
  const float2* tw;      // exp(-2 pi i k / twn), k < twn / 2
  int twn;
};

// per-pair state: everything lives in the caller's workspace
struct Pair {
  float* data[2];        // [ref, deg]: SIGNAL_INFO.data (model signals), NA floats each
  float* adata[2];       // alignment copies (DC-blocked, IIR-filtered)
  float* vad[2];
  float* logvad[2];
  int nsamp[2];          // Nsamples incl. the two search buffers
  int na;                // allocated floats per signal
  float2 *ca, *cb;       // FFT ping-pong, p2max complex each
  int p2max;
  float* scratch;        // >= 8 * NW + 8 * 1024 floats
  float *ppd_ref, *ppd_deg, *fd, *fda, *tpr, *tweaked, *doubly;
  int* st;               // integer state (Err), see indices below
  float* fst;            // float state
};

// integer state layout
enum { I_NUTT = 0, I_CRUDE = 1, I_SSTART = 2, I_SEND = I_SSTART + MAXNUTT, I_DEST = I_SEND + MAXNUTT, I_DELAY = I_DEST + MAXNUTT,
       I_START = I_DELAY + MAXNUTT, I_END = I_START + MAXNUTT, I_TMP = I_END + MAXNUTT, I_COUNT = I_TMP + 64 };
enum { F_CONF = 0, F_TMP = MAXNUTT, F_COUNT = F_TMP + 64 };

// stage timer for the trace (microseconds of the 100 MHz wall clock on the device, 0 on the host)
PQ_FN inline long long stamp() {
#if PQ_DEVICE
  return (long long)wall_clock64();
#else
  return 0;
#endif
}

PQ_FN inline int nextpow2(int x) {
  int n = 1;
  while (n < x) n <<= 1;
  return n;
}
PQ_FN inline int cdiv(int a, int b) { return a / b; }      // C division truncates towards zero: what the standard's code does

struct Lds {                 // team-shared (LDS) buffers: frame transforms (1024-point max), VAD scan copy, IIR hand-off
  float2 *la, *lb;
  float *x, *h;
  float* iir;                // 512 floats
  float* w;                  // wcap floats
  int wcap;
};

// ---- FFT: Stockham autosort, radix-4 passes (+ one radix-2 pass when log2 n is odd), out of place between a and b; returns the
// buffer that holds the result.  `batch` independent n-point transforms stored back to back share the passes (and barriers).
PQ_FN inline float2 tw_at(const Params& P, int idx) {      // exp(-2 pi i idx / twn), idx < twn
  const int h = P.twn >> 1;
  float2 w = P.tw[idx < h ? idx : idx - h];
  if (idx >= h) { w.x = -w.x; w.y = -w.y; }
  return w;
}
PQ_FN inline float2 cmul(float2 a, float2 w) {
  float2 r;
  r.x = a.x * w.x - a.y * w.y; r.y = a.x * w.y + a.y * w.x;
  return r;
}

PQ_FN inline float2* fft_batched(const Team& T, float2* a, float2* b, int n, int batch, bool inverse, const Params& P) {
  const int step = P.twn / n;
  float2 *x = a, *y = b;
  int nc = n, st = 1;                                      // current sub-length, stride
  while (nc >= 4) {
    const int n1 = nc >> 2, per = n1 * st;                 // butterflies per transform in this pass = n / 4
    for (int id = T.tid; id < per * batch; id += T.nt) {
      const int q0 = id / per, r = id - q0 * per;
      const int p = r / st, q = r - p * st;
      float2* xb = x + q0 * n;
      float2* yb = y + q0 * n;
      const float2 va = xb[q + st * p], vb = xb[q + st * (p + n1)], vc = xb[q + st * (p + 2 * n1)], vd = xb[q + st * (p + 3 * n1)];
      float2 apc, amc, bpd, jbmd;
      apc.x = va.x + vc.x; apc.y = va.y + vc.y;
      amc.x = va.x - vc.x; amc.y = va.y - vc.y;
      bpd.x = vb.x + vd.x; bpd.y = vb.y + vd.y;
      const float dx = vb.x - vd.x, dy = vb.y - vd.y;
      if (!inverse) { jbmd.x = -dy; jbmd.y = dx; } else { jbmd.x = dy; jbmd.y = -dx; }     // (+-i) (b - d)
      float2 w1 = tw_at(P, p * st * step), w2 = tw_at(P, 2 * p * st * step), w3 = tw_at(P, 3 * p * st * step);
      if (inverse) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
      float2 o0, t1, t2, t3;
      o0.x = apc.x + bpd.x; o0.y = apc.y + bpd.y;
      t1.x = amc.x - jbmd.x; t1.y = amc.y - jbmd.y;
      t2.x = apc.x - bpd.x; t2.y = apc.y - bpd.y;
      t3.x = amc.x + jbmd.x; t3.y = amc.y + jbmd.y;
      yb[q + st * (4 * p)] = o0;
      yb[q + st * (4 * p + 1)] = cmul(t1, w1);
      yb[q + st * (4 * p + 2)] = cmul(t2, w2);
      yb[q + st * (4 * p + 3)] = cmul(t3, w3);
    }
    T.sync();
    float2* t = x; x = y; y = t;
    nc >>= 2; st <<= 2;
  }
  if (nc == 2) {                                            // last pass of an odd power of two: plain butterflies, stride n / 2
    const int per = st;
    for (int id = T.tid; id < per * batch; id += T.nt) {
      const int q0 = id / per, q = id - q0 * per;
      float2* xb = x + q0 * n;
      float2* yb = y + q0 * n;
      const float2 c0 = xb[q], c1 = xb[q + st];
      float2 s0, s1;
      s0.x = c0.x + c1.x; s0.y = c0.y + c1.y;
      s1.x = c0.x - c1.x; s1.y = c0.y - c1.y;
      yb[q] = s0; yb[q + st] = s1;
    }
    T.sync();
    float2* t = x; x = y; y = t;
  }
  return x;
}

PQ_FN inline float2* fft(const Team& T, float2* a, float2* b, int n, bool inverse, const Params& P) {
  return fft_batched(T, a, b, n, 1, inverse, P);
}

// Long transforms (2^13 .. 2^17 points) as N = N1 x 1024 (four-step): N1-point transforms down the columns (1024 / N1 columns
// per LDS tile), twiddle, 1024-point transforms along the rows - the signal crosses global memory twice instead of once per
// radix-2 pass.  Forward: natural order in `src`, PERMUTED order out in `dst`: dst[pos(k)] = X[k], pos(k) = (k mod N1) *
// 1024 + k / N1.  Inverse: permuted order in, natural order out (not scaled).  la / lb: 1024-point LDS buffers.
PQ_FN inline int big_pos(int k, int n1) { return (k & (n1 - 1)) * 1024 + k / n1; }

PQ_FN inline void fft_big_forward(const Team& T, const Params& P, float2* src, float2* dst, int n, float2* la, float2* lb) {
  const int n1 = n >> 10, cols = 1024 / n1, tstep = P.twn / n;
  for (int c0 = 0; c0 < 1024; c0 += cols) {            // columns n2 = c0 .. c0 + cols - 1
    for (int i = T.tid; i < 1024; i += T.nt) {
      const int r = i / cols, c = i - r * cols;          // element (n1 index r, column c): coalesced over c
      la[c * n1 + r] = src[1024 * r + c0 + c];
    }
    T.sync();
    float2* R = fft_batched(T, la, lb, n1, cols, false, P);
    for (int i = T.tid; i < 1024; i += T.nt) {
      const int k1 = i / cols, c = i - k1 * cols, n2 = c0 + c;
      const float2 v = R[c * n1 + k1];
      const int e = n2 * k1;                              // < 1024 * N1 = n: no reduction needed
      float2 w = P.tw[(e < n / 2 ? e : e - n / 2) * tstep];
      if (e >= n / 2) { w.x = -w.x; w.y = -w.y; }
      float2 o;
      o.x = v.x * w.x - v.y * w.y; o.y = v.x * w.y + v.y * w.x;
      dst[k1 * 1024 + n2] = o;
    }
    T.sync();
  }
  for (int k1 = 0; k1 < n1; ++k1) {                       // rows: 1024-point transforms in place
    for (int i = T.tid; i < 1024; i += T.nt) la[i] = dst[k1 * 1024 + i];
    T.sync();
    float2* R = fft(T, la, lb, 1024, false, P);
    for (int i = T.tid; i < 1024; i += T.nt) dst[k1 * 1024 + i] = R[i];
    T.sync();
  }
}

PQ_FN inline void fft_big_inverse(const Team& T, const Params& P, float2* src, float2* dst, int n, float2* la, float2* lb) {
  const int n1 = n >> 10, cols = 1024 / n1, tstep = P.twn / n;
  for (int k1 = 0; k1 < n1; ++k1) {
    for (int i = T.tid; i < 1024; i += T.nt) la[i] = src[k1 * 1024 + i];
    T.sync();
    float2* R = fft(T, la, lb, 1024, true, P);
    for (int n2 = T.tid; n2 < 1024; n2 += T.nt) {
      const float2 v = R[n2];
      const int e = n2 * k1;                              // < 1024 * N1 = n: no reduction needed
      float2 w = P.tw[(e < n / 2 ? e : e - n / 2) * tstep];
      if (e >= n / 2) { w.x = -w.x; w.y = -w.y; }
      float2 o;                                           // times conj(w)
      o.x = v.x * w.x + v.y * w.y; o.y = v.y * w.x - v.x * w.y;
      src[k1 * 1024 + n2] = o;
    }
    T.sync();
  }
  for (int c0 = 0; c0 < 1024; c0 += cols) {
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
    T.sync();
  }
}

PQ_FN inline float interpolate(float freq, const float (*curve)[2], int n) {
  int lo, hi;
  if (freq <= curve[0][0]) { lo = 0; hi = 1; }
  else if (freq >= curve[n - 1][0]) { lo = n - 2; hi = n - 1; }
  else {
    hi = 1;
    while (curve[hi][0] < freq) ++hi;
    lo = hi - 1;
  }
  const double fl = curve[lo][0], fh = curve[hi][0], gl = curve[lo][1], gh = curve[hi][1];
  return (float)((((double)freq - fl) * gh + (fh - (double)freq) * gl) / (fh - fl));
}

// apply_filter: FFT-domain filter of data[sb .. sb + n), gain relative to 1 kHz
PQ_FN inline void apply_filter(const Team& T, const Params& P, Pair& S, const Lds& L, float* data, int nsamples,
                               const float (*curve)[2], int npts) {
  const int sb = SEARCHBUFFER * P.ds;
  const int n = nsamples - 2 * sb + P.pad;
  int p2 = nextpow2(n);
  if (p2 < 2048) p2 = 2048;                               // (four-step form needs N1 >= 2; a longer transform filters the same)
  for (int i = T.tid; i < p2; i += T.nt) {
    float2 v;
    v.x = i < n ? data[sb + i] : 0.f;
    v.y = 0.f;
    S.ca[i] = v;
  }
  T.sync();
  fft_big_forward(T, P, S.ca, S.cb, p2, L.la, L.lb);
  const float ref_gain = interpolate(1000.f, curve, npts);
  const float res = (float)P.fs / (float)p2;
  const int n1 = p2 >> 10;
  for (int i = T.tid; i <= p2 / 2; i += T.nt) {
    const float db = interpolate(i * res, curve, npts) - ref_gain;
    const float fac = powf(10.f, db / 20.f);
    const int a = big_pos(i, n1);
    S.cb[a].x *= fac; S.cb[a].y *= fac;
    if (i > 0 && i < p2 / 2) { const int b = big_pos(p2 - i, n1); S.cb[b].x *= fac; S.cb[b].y *= fac; }
  }
  T.sync();
  fft_big_inverse(T, P, S.cb, S.ca, p2, L.la, L.lb);
  const float inv = 1.f / (float)p2;
  for (int i = T.tid; i < n; i += T.nt) data[sb + i] = S.ca[i].x * inv;
  T.sync();
}

PQ_FN inline double pow_of(const Team& T, const float* x, int start, int stop, int divisor) {
  double p = 0;
  for (int i = start + T.tid; i < stop; i += T.nt) p += (double)x[i] * (double)x[i];
  return T.sum(p) / divisor;
}

// cascade of direct-form-II sections {b0, b1, b2, a1, a2} over xa[0, n) and xb[0, n) (the two signals of the pair at once).
// Device: one wave; signal a lives on lanes 0-15, signal b on lanes 16-31, section s on lane s of its row, and a sample moves
// from section to section with a one-lane DPP row shift (a few cycles; the generic shuffle is a 100-cycle LDS permute, and the
// recurrence makes every iteration wait for it).  Inputs are fetched 64 at a time by the whole wave and handed out with
// v_readlane; the last section's outputs collect in LDS and leave 64 at a time.  Host: plain loops.
PQ_FN inline void iir_cascade2(const Team& T, float* xa, float* xb, int n, const float (*sos)[5], int nsos, float* obuf) {
#if PQ_DEVICE
  if (T.tid < 64) {
    const int lane = T.tid, row = lane >> 4, sec = lane & 15;
    const bool mine = row < 2 && sec < nsos;
    // sections that do not exist (and the idle rows) carry zero coefficients: their arithmetic runs unconditionally and yields 0
    float b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
    if (mine) { b0 = sos[sec][0]; b1 = sos[sec][1]; b2 = sos[sec][2]; a1 = sos[sec][3]; a2 = sos[sec][4]; }
    const bool first = sec == 0, rowa = row == 0, last = mine && sec == nsos - 1;
    // 256-entry ring of the last section's outputs per signal, in LDS.  (A volatile generic pointer here compiled to a
    // write-through FLAT store followed by s_waitcnt vmcnt(0) in EVERY iteration: 160 cycles per sample.)  One wave writes and
    // reads it in program order; the fences below only keep the compiler from moving the accesses.
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float* ring = (lds_float*)obuf + (row & 1) * 256;
    lds_float* ring_all = (lds_float*)obuf;
    float z1 = 0.f, z2 = 0.f, out_prev = 0.f;
    const int total = n + nsos - 1;
    float ca = lane < n ? xa[lane] : 0.f, cb = lane < n ? xb[lane] : 0.f;
    for (int base = 0; base < total; base += 64) {
      const float na = (base + 64 + lane < n) ? xa[base + 64 + lane] : 0.f;     // next 64 inputs, in flight under the loop
      const float nb = (base + 64 + lane < n) ? xb[base + 64 + lane] : 0.f;
      // Before its first sample a section sees zeros (state stays 0); past the end it computes values nobody reads: no branches.
#pragma unroll 8
      for (int k = 0; k < 64; ++k) {
        const float ina = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ca), k));     // (the builtin moves ints: bit casts,
        const float inb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb), k));     //  not value conversions)
        const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(out_prev), 0x111, 0xf, 0xf, true));  // row_shr:1
        const float in = first ? (rowa ? ina : inb) : up;
        // transposed direct form II: one multiply-add between a section's input and its output (the samples' way down the
        // cascade is the serial chain of this loop); the standard's code uses direct form II - same filter, rounding differs
        const float out = b0 * in + z1;
        z1 = b1 * in - a1 * out + z2;
        z2 = b2 * in - a2 * out;
        if (last) ring[(base + k - (nsos - 1)) & 255] = out;
        out_prev = out;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int si = base - (nsos - 1) + lane;                 // outputs completed during this chunk
      if (si >= 0 && si < n) { xa[si] = ring_all[si & 255]; xb[si] = ring_all[256 + (si & 255)]; }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      ca = na; cb = nb;
    }
  }
  T.sync();
#else
  (void)obuf;
  for (int q = 0; q < 2; ++q) {
    float* x = q == 0 ? xa : xb;
    for (int s = 0; s < nsos; ++s) {
      const float b0 = sos[s][0], b1 = sos[s][1], b2 = sos[s][2], a1 = sos[s][3], a2 = sos[s][4];
      float z1 = 0.f, z2 = 0.f;
      for (int i = 0; i < n; ++i) {
        const float z0 = x[i] - a1 * z1 - a2 * z2;
        x[i] = b0 * z0 + b1 * z1 + b2 * z2;
        z2 = z1; z1 = z0;
      }
    }
  }
  (void)T;
#endif
}

PQ_FN inline void fix_power_level(const Team& T, const Params& P, Pair& S, const Lds& L, int sig, int maxn) {
  const int sb = SEARCHBUFFER * P.ds, n = S.nsamp[sig];
  float* tmp = S.tweaked;                                  // free at this stage
  for (int i = T.tid; i < n + P.pad; i += T.nt) tmp[i] = S.data[sig][i];
  T.sync();
  apply_filter(T, P, S, L, tmp, n, ALIGN_FILTER_DB, 26);
  const double p = pow_of(T, tmp, sb, n - sb + P.pad, maxn - 2 * sb + P.pad);
  const float g = p > 0.0 ? (float)sqrt(1.0e7 / p) : 1.0f;          // (an all-zero signal is left alone)
  for (int i = T.tid; i < n; i += T.nt) S.data[sig][i] *= g;
  T.sync();
}

PQ_FN inline void dc_block(const Team& T, const Params& P, float* data, int nsamples) {
  const int ofs = SEARCHBUFFER * P.ds, cnt = nsamples - 2 * ofs;
  double acc = 0;
  for (int i = T.tid; i < cnt; i += T.nt) acc += data[ofs + i];
  const float mean = (float)(T.sum(acc) / nsamples);
  for (int i = T.tid; i < cnt; i += T.nt) data[ofs + i] -= mean;
  T.sync();
  for (int i = T.tid; i < P.ds; i += T.nt) {
    const float r = (0.5f + i) / P.ds;
    data[ofs + i] *= r;
    data[nsamples - ofs - 1 - i] *= r;
  }
  T.sync();
}

// apply_VAD: window energies in parallel, threshold iteration with block reductions, the scans by thread 0
PQ_FN inline void apply_vad(const Team& T, const Params& P, const float* data, int nsamples, float* vad_g, float* logvad, const Lds& L) {
  const int ds = P.ds, nw = nsamples / ds;
  float* vad = nw <= L.wcap ? L.w : vad_g;          // the scans of thread 0 run on an LDS copy when it fits
  double s = 0;
  float mx = 0.f;
  for (int w = T.tid; w < nw; w += T.nt) {
    float e = 0.f;
    for (int k = 0; k < ds; ++k) { const float g = data[w * ds + k]; e += g * g; }
    e /= ds;
    vad[w] = e;
    s += e;
    mx = fmaxf(mx, e);
  }
  float level_thresh = (float)(T.sum(s) / nw);
  float level_min = T.maxf(mx);
  level_min = level_min > 0.f ? level_min * 1.0e-4f : 1.0f;
  for (int w = T.tid; w < nw; w += T.nt) if (vad[w] < level_min) vad[w] = level_min;
  T.sync();
  float level_noise = 0.f, std_noise = 0.f;
  for (int it = 0; it < 12; ++it) {
    double a = 0, c = 0;
    for (int w = T.tid; w < nw; w += T.nt) if (vad[w] <= level_thresh) { a += vad[w]; c += 1; }
    a = T.sum(a); c = T.sum(c);
    level_noise = 0.f; std_noise = 0.f;
    if (c > 0) {
      level_noise = (float)(a / c);
      double q = 0;
      for (int w = T.tid; w < nw; w += T.nt) if (vad[w] <= level_thresh) { const double g = vad[w] - level_noise; q += g * g; }
      std_noise = (float)sqrt(T.sum(q) / c);
    }
    level_thresh = 1.001f * (level_noise + 2.0f * std_noise);
  }
  double sig = 0, noi = 0, len = 0;
  for (int w = T.tid; w < nw; w += T.nt) {
    if (vad[w] > level_thresh) { sig += vad[w]; len += 1; } else noi += vad[w];
  }
  sig = T.sum(sig); noi = T.sum(noi); len = T.sum(len);
  float level_sig = len > 0 ? (float)(sig / len) : 0.f;
  if (len == 0) level_thresh = -1.0f;
  level_noise = len < nw ? (float)(noi / (nw - len)) : 1.0f;
  for (int w = T.tid; w < nw; w += T.nt) if (vad[w] <= level_thresh) vad[w] = -vad[w];
  T.sync();
  if (T.tid == 0) {
    vad[0] = -level_min; vad[nw - 1] = -level_min;
    int start = 0, finish = 0;
    for (int i = 1; i < nw; ++i) {
      if (vad[i] > 0.f && vad[i - 1] <= 0.f) start = i;
      if (vad[i] <= 0.f && vad[i - 1] > 0.f) {
        finish = i;
        if (finish - start <= MINSPEECHLGTH) for (int k = start; k < finish; ++k) vad[k] = -vad[k];
      }
    }
    if (level_sig >= level_noise * 1000.0f) {
      for (int i = 1; i < nw; ++i) {
        if (vad[i] > 0.f && vad[i - 1] <= 0.f) start = i;
        if (vad[i] <= 0.f && vad[i - 1] > 0.f) {
          finish = i;
          float g = 0.f;
          for (int k = start; k < finish; ++k) g += vad[k];
          if (g < 3.0f * level_thresh * (finish - start)) for (int k = start; k < finish; ++k) vad[k] = -vad[k];
        }
      }
    }
    start = 0; finish = 0;
    for (int i = 1; i < nw; ++i) {
      if (vad[i] > 0.f && vad[i - 1] <= 0.f) {
        start = i;
        if (finish > 0 && start - finish <= JOINSPEECHLGTH) for (int k = finish; k < start; ++k) vad[k] = level_min;
      }
      if (vad[i] <= 0.f && vad[i - 1] > 0.f) finish = i;
    }
    start = 0;
    for (int i = 1; i < nw; ++i) if (vad[i] > 0.f && vad[i - 1] <= 0.f) start = i;
    if (start == 0) {
      for (int i = 0; i < nw; ++i) vad[i] = fabsf(vad[i]);
      vad[0] = -level_min; vad[nw - 1] = -level_min;
    }
    int i = 3;
    while (i < nw - 2) {
      if (vad[i] > 0.f && vad[i - 2] <= 0.f) { vad[i - 2] = vad[i] * 0.1f; vad[i - 1] = vad[i] * 0.3f; ++i; }
      if (vad[i] <= 0.f && vad[i - 1] > 0.f) { vad[i] = vad[i - 1] * 0.3f; vad[i + 1] = vad[i - 1] * 0.1f; i += 3; }
      ++i;
    }
  }
  T.sync();
  if (level_thresh <= 0.f) level_thresh = level_min;
  for (int w = T.tid; w < nw; w += T.nt) {
    float v = vad[w];
    if (v < 0.f) v = 0.f;
    vad_g[w] = v;
    logvad[w] = v <= level_thresh ? 0.f : logf(v / level_thresh);
  }
  T.sync();
}

// crude_align: lag of the maximum of the cross-correlation of the log-VAD envelopes, computed directly (the envelopes are
// ~1,200 values: 1.5 M multiply-adds per pair, less than one long FFT)
PQ_FN inline void crude_align(const Team& T, const Params& P, Pair& S, int utt_id) {
  const int ds = P.ds;
  int* st = S.st;
  const int nd_all = S.nsamp[1] / ds;
  int nr, nd, startr, startd;
  if (utt_id == -1) { nr = S.nsamp[0] / ds; nd = nd_all; startr = 0; startd = 0; }
  else {
    const int k = utt_id == MAXNUTT ? MAXNUTT - 1 : utt_id;
    const int est = utt_id == MAXNUTT ? st[I_DEST + MAXNUTT - 1] : st[I_CRUDE];
    startr = st[I_SSTART + k];
    startd = startr + cdiv(est, ds);
    if (startd < 0) { startr = cdiv(-est, ds); startd = 0; }
    nr = st[I_SEND + k] - startr;
    nd = nr;
    if (startd + nd > nd_all) nd = nd_all - startd;
  }
  float best = 0.f;
  int bidx = -1;
  if (nr > 1 && nd > 1) {
    const float* x1 = S.logvad[0] + startr;
    const float* x2 = S.logvad[1] + startd;
    float lb = 0.f;
    int li = -1;
    for (int k = T.tid; k < nr + nd - 1; k += T.nt) {
      const int lag = k - (nr - 1);
      int i0 = lag < 0 ? -lag : 0, i1 = nr;
      if (i1 > nd - lag) i1 = nd - lag;
      float acc = 0.f;
      for (int i = i0; i < i1; ++i) acc += x1[i] * x2[i + lag];
      if (acc > lb) { lb = acc; li = k; }
    }
    T.argmax(lb, li, &best, &bidx);
  }
  const int i_max = (bidx >= 0 && best > 0.f) ? bidx : nr - 1;
  const int lag = (i_max - nr + 1) * ds;
  if (T.tid == 0) {
    if (utt_id == -1) st[I_CRUDE] = lag;
    else if (utt_id == MAXNUTT) st[I_DELAY + MAXNUTT - 1] = lag + st[I_DEST + MAXNUTT - 1];
    else st[I_DEST + utt_id] = lag + st[I_CRUDE];
  }
  T.sync();
}

// the VAD scan shared by id_searchwindows (search = true) and id_utterances (thread 0)
PQ_FN inline int utt_scan(const Params& P, Pair& S, bool search) {
  const int ds = P.ds;
  int* st = S.st;
  const float* vad = S.vad[0];
  const int n = S.nsamp[0] / ds;
  const int del_deg_start = MINUTTLENGTH - cdiv(st[I_CRUDE], ds);
  const int del_deg_end = cdiv(S.nsamp[1] - st[I_CRUDE], ds) - MINUTTLENGTH;
  int num = 0, flag = 0, this_start = 0;
  for (int i = 0; i < n; ++i) {
    const float v = vad[i];
    if (v > 0.f && flag == 0) {
      flag = 1; this_start = i;
      if (search) { int s = i - SEARCHBUFFER; st[I_SSTART + num] = s < 0 ? 0 : s; } else st[I_START + num] = i;
    }
    if ((v == 0.f || i == n - 1) && flag == 1) {
      flag = 0;
      if (search) { int e = i + SEARCHBUFFER; st[I_SEND + num] = e > n - 1 ? n - 1 : e; } else st[I_END + num] = i;
      if (i - this_start >= MINUTTLENGTH && this_start < del_deg_end && i > del_deg_start) {
        ++num;
        if (num >= MAXNUTT - 1) break;
      }
    }
  }
  return num;
}

// one Hann-windowed frame pair -> |cross-correlation| in lds_x (n floats); returns 0.99 * max
PQ_FN inline float frame_xcorr(const Team& T, const Params& P, Pair& S, int startr, int startd, float2* la, float2* lb, float* lds_x) {
  const int n = P.align_nfft;
  // one complex FFT carries both real frames: z = ref + i deg
  for (int i = T.tid; i < n; i += T.nt) {
    const float w = 0.5f * (1.0f - cosf(TWOPI_F * i / n));
    float2 v;
    v.x = S.adata[0][startr + i] * w;
    v.y = S.adata[1][startd + i] * w;
    la[i] = v;
  }
  T.sync();
  float2* Z = fft(T, la, lb, n, false, P);
  float2* O = Z == la ? lb : la;
  // X1 = (Z[k] + conj Z[n-k]) / 2, X2 = (Z[k] - conj Z[n-k]) / (2i); product conj(X1) X2
  for (int k = T.tid; k < n; k += T.nt) {
    const float2 a = Z[k], b = Z[(n - k) & (n - 1)];
    const float x1r = 0.5f * (a.x + b.x), x1i = 0.5f * (a.y - b.y);
    const float x2r = 0.5f * (a.y + b.y), x2i = -0.5f * (a.x - b.x);
    float2 p;
    p.x = x1r * x2r + x1i * x2i;
    p.y = x1r * x2i - x1i * x2r;
    O[k] = p;
  }
  T.sync();
  float2* R = fft(T, O, Z, n, true, P);
  float mx = 0.f;
  const float inv = 1.f / n;
  for (int i = T.tid; i < n; i += T.nt) {
    const float v = fabsf(R[i].x * inv);
    lds_x[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = T.maxf(mx);
  return mx * 0.99f;
}

PQ_FN inline void hist_peak(const Team& T, int n, const float* h, float hsum, int est, int* delay, float* conf) {
  float lb = 0.f;
  int li = -1;
  for (int i = T.tid; i < n; i += T.nt) if (h[i] > lb) { lb = h[i]; li = i; }
  float best;
  int bidx;
  T.argmax(lb, li, &best, &bidx);
  int i_max = (bidx >= 0 && best > 0.f) ? bidx : 0;
  if (!(bidx >= 0 && best > 0.f)) best = 0.f;
  if (i_max >= n / 2) i_max -= n;
  *delay = est + i_max;
  *conf = hsum > 0.f ? best / hsum : 0.f;
}

PQ_FN inline void time_align(const Team& T, const Params& P, Pair& S, const Lds& L, int utt_id) {
  const int n = P.align_nfft, ds = P.ds;
  int* st = S.st;
  const int est = st[I_DEST + utt_id];
  for (int i = T.tid; i < n; i += T.nt) L.h[i] = 0.f;
  T.sync();
  int startr = st[I_SSTART + utt_id] * ds, startd = startr + est;
  if (startd < 0) { startr = -est; startd = 0; }
  while (startd + n <= S.nsamp[1] && startr + n / 4 <= st[I_SEND + utt_id] * ds) {
    const float v_max = frame_xcorr(T, P, S, startr, startd, L.la, L.lb, L.x);
    const float add = powf(v_max, 0.125f);
    for (int i = T.tid; i < n; i += T.nt) if (L.x[i] > v_max) L.h[i] += add;
    T.sync();
    startr += n / 4; startd += n / 4;
  }
  double hs = 0;
  for (int i = T.tid; i < n; i += T.nt) hs += L.h[i];
  const float hsum = (float)T.sum(hs);
  // smooth the histogram with the triangular kernel (circular), directly: 2 * kernel - 1 taps
  const int kernel = n / 64;
  for (int i = T.tid; i < n; i += T.nt) {
    float acc = L.h[i];
    for (int k = 1; k < kernel; ++k) {
      const float w = 1.0f - (float)k / (float)kernel;
      acc += w * (L.h[(i + k) & (n - 1)] + L.h[(i - k + n) & (n - 1)]);
    }
    L.x[i] = fabsf(acc);
  }
  T.sync();
  int delay;
  float conf;
  hist_peak(T, n, L.x, hsum, est, &delay, &conf);
  if (T.tid == 0) { st[I_DELAY + utt_id] = delay; S.fst[F_CONF + utt_id] = conf; }
  T.sync();
}

PQ_FN inline void id_utterances(const Params& P, Pair& S) {           // thread 0
  const int ds = P.ds;
  int* st = S.st;
  utt_scan(P, S, false);
  const int n = S.nsamp[0] / ds, nu = st[I_NUTT];
  st[I_START] = SEARCHBUFFER;
  st[I_END + nu - 1] = n - SEARCHBUFFER;
  for (int k = 1; k < nu; ++k) {
    const int mid = (st[I_START + k] + st[I_END + k - 1]) / 2;
    st[I_START + k] = mid; st[I_END + k - 1] = mid;
  }
  int this_start = st[I_START] * ds + st[I_DELAY];
  if (this_start < SEARCHBUFFER * ds) st[I_START] = SEARCHBUFFER + cdiv(ds - 1 - st[I_DELAY], ds);
  int last_end = st[I_END + nu - 1] * ds + st[I_DELAY + nu - 1];
  if (last_end > S.nsamp[1] - SEARCHBUFFER * ds) st[I_END + nu - 1] = cdiv(S.nsamp[1] - st[I_DELAY + nu - 1], ds) - SEARCHBUFFER;
  for (int k = 1; k < nu; ++k) {
    this_start = st[I_START + k] * ds + st[I_DELAY + k];
    last_end = st[I_END + k - 1] * ds + st[I_DELAY + k - 1];
    if (this_start < last_end) {
      const int mid = cdiv(this_start + last_end, 2);
      st[I_START + k] = cdiv(ds - 1 + mid - st[I_DELAY + k], ds);
      st[I_END + k - 1] = cdiv(mid - st[I_DELAY + k - 1], ds);
    }
  }
}

// split_align: try to split one utterance at up to 40 break points; the winner (if any) goes to tmp state
struct Split { int ed1, d1, ed2, d2, bp; float dc1, dc2; };

PQ_FN inline void split_accumulate(const Team& T, const Params& P, Pair& S, const Lds& L, int startr, int startd, float* hsum) {
  const int n = P.align_nfft, kernel = n / 64;
  const float v_max = frame_xcorr(T, P, S, startr, startd, L.la, L.lb, L.x);
  const float n_max = powf(v_max, 0.125f) / kernel;
  // H[c + k] += n_max * (kernel - |k|) for every c above the threshold: gather form, one writer per bin
  float cnt = 0.f;
  for (int i = T.tid; i < n; i += T.nt) {
    float acc = 0.f;
    for (int k = 1 - kernel; k < kernel; ++k) {
      const int c = (i - k + n) & (n - 1);
      if (L.x[c] > v_max) acc += n_max * (float)(kernel - (k < 0 ? -k : k));
    }
    L.h[i] += acc;
    if (L.x[i] > v_max) cnt += 1.f;
  }
  *hsum += (float)T.sum(cnt) * n_max * kernel;
}

PQ_FN inline Split split_align(const Team& T, const Params& P, Pair& S, const Lds& L, int utt_start, int speech_start,
                               int speech_end, int utt_end, int delay_est, float delay_conf) {
  const int n = P.align_nfft, ds = P.ds;
  int* st = S.st;
  int* bps = st + I_TMP;                    // up to 41 break points
  int* ed1 = reinterpret_cast<int*>(S.scratch);        // 4 x 41 ints + 2 x 41 floats in the scratch area
  int* ed2 = ed1 + 41;
  int* d1 = ed2 + 41;
  int* d2 = d1 + 41;
  float* dc1 = reinterpret_cast<float*>(d2 + 41);
  float* dc2 = dc1 + 41;
  const int utt_len = speech_end - speech_start, test = MAXNUTT - 1;
  const int delta = n / (4 * ds);
  int step = (int)((0.801 * utt_len + 40 * delta - 1) / (40 * delta));
  step *= delta;
  int pad = utt_len / 10;
  if (pad < 75) pad = 75;
  int nbp = 0;
  if (T.tid == 0) {
    bps[0] = speech_start + pad;
    do { ++nbp; bps[nbp] = bps[nbp - 1] + step; } while (bps[nbp] <= speech_end - pad && nbp < 40);
    st[I_TMP + 63] = nbp;
  }
  T.sync();
  nbp = st[I_TMP + 63];
  Split best;
  best.dc1 = 0.f; best.dc2 = 0.f; best.ed1 = best.d1 = best.ed2 = best.d2 = best.bp = 0;
  if (nbp <= 0) return best;
  for (int bp = 0; bp < nbp; ++bp) {
    if (T.tid == 0) { st[I_DEST + test] = delay_est; st[I_SSTART + test] = utt_start; st[I_SEND + test] = bps[bp]; }
    T.sync();
    crude_align(T, P, S, MAXNUTT);
    if (T.tid == 0) { ed1[bp] = st[I_DELAY + test]; st[I_DEST + test] = delay_est; st[I_SSTART + test] = bps[bp]; st[I_SEND + test] = utt_end; }
    T.sync();
    crude_align(T, P, S, MAXNUTT);
    if (T.tid == 0) { ed2[bp] = st[I_DELAY + test]; dc1[bp] = -2.0f; }
    T.sync();
  }
  while (true) {
    int bp = 0;
    while (bp < nbp && dc1[bp] > -2.0f) ++bp;
    if (bp >= nbp) break;
    const int est = ed1[bp];
    for (int i = T.tid; i < n; i += T.nt) L.h[i] = 0.f;
    T.sync();
    float hsum = 0.f;
    int startr = utt_start * ds, startd = startr + est;
    if (startd < 0) { startr = -est; startd = 0; }
    while (true) {
      while (startd + n <= S.nsamp[1] && startr + n / 4 <= bps[bp] * ds) {
        split_accumulate(T, P, S, L, startr, startd, &hsum);
        T.sync();
        startr += n / 4; startd += n / 4;
      }
      int dl;
      float cf;
      hist_peak(T, n, L.h, hsum, est, &dl, &cf);
      if (T.tid == 0) { d1[bp] = dl; dc1[bp] = cf; }
      T.sync();
      int nxt = -1;
      while (bp < nbp - 1) {
        ++bp;
        if (ed1[bp] == est && dc1[bp] <= -2.0f) { nxt = bp; break; }
      }
      if (nxt < 0) break;
    }
  }
  if (T.tid == 0) for (int bp = 0; bp < nbp; ++bp) dc2[bp] = dc1[bp] > delay_conf ? -2.0f : 0.0f;
  T.sync();
  while (true) {
    int bp = nbp - 1;
    while (bp >= 0 && dc2[bp] > -2.0f) --bp;
    if (bp < 0) break;
    const int est = ed2[bp];
    for (int i = T.tid; i < n; i += T.nt) L.h[i] = 0.f;
    T.sync();
    float hsum = 0.f;
    int startr = utt_end * ds - n, startd = startr + est;
    if (startd + n > S.nsamp[1]) { startd = S.nsamp[1] - n; startr = startd - est; }
    while (true) {
      while (startd >= 0 && startr + n * 3 / 4 >= bps[bp] * ds) {
        split_accumulate(T, P, S, L, startr, startd, &hsum);
        T.sync();
        startr -= n / 4; startd -= n / 4;
      }
      int dl;
      float cf;
      hist_peak(T, n, L.h, hsum, est, &dl, &cf);
      if (T.tid == 0) { d2[bp] = dl; dc2[bp] = cf; }
      T.sync();
      int nxt = -1;
      while (bp > 0) {
        --bp;
        if (ed2[bp] == est && dc2[bp] <= -2.0f) { nxt = bp; break; }
      }
      if (nxt < 0) break;
    }
  }
  for (int bp = 0; bp < nbp; ++bp) {
    const int diff = d2[bp] - d1[bp];
    if ((diff < 0 ? -diff : diff) >= ds && dc1[bp] + dc2[bp] > best.dc1 + best.dc2 && dc1[bp] > delay_conf && dc2[bp] > delay_conf) {
      best.ed1 = ed1[bp]; best.d1 = d1[bp]; best.dc1 = dc1[bp];
      best.ed2 = ed2[bp]; best.d2 = d2[bp]; best.dc2 = dc2[bp];
      best.bp = bps[bp];
    }
  }
  T.sync();
  return best;
}

PQ_FN inline void utterance_split(const Team& T, const Params& P, Pair& S, const Lds& L) {
  const int ds = P.ds;
  int* st = S.st;
  float* conf = S.fst + F_CONF;
  const float* vad = S.vad[0];
  int k = 0;
  while (k < st[I_NUTT] && st[I_NUTT] < MAXNUTT) {
    const int us = st[I_START + k], ue = st[I_END + k];
    const float cf = conf[k];
    const int dest = st[I_DEST + k];
    int ss = us;
    while (ss < ue && vad[ss] <= 0.f) ++ss;
    int se = ue;
    while (se > us && vad[se] <= 0.f) --se;
    ++se;
    bool split = false;
    if (se - ss >= 200) {
      const Split b = split_align(T, P, S, L, us, ss, se, ue, dest, cf);
      if (b.dc1 > cf && b.dc2 > cf) {
        split = true;
        if (T.tid == 0) {
          for (int s = st[I_NUTT] - 1; s > k; --s) {
            st[I_DEST + s + 1] = st[I_DEST + s]; st[I_DELAY + s + 1] = st[I_DELAY + s]; conf[s + 1] = conf[s];
            st[I_START + s + 1] = st[I_START + s]; st[I_END + s + 1] = st[I_END + s];
            st[I_SSTART + s + 1] = st[I_START + s]; st[I_SEND + s + 1] = st[I_END + s];
          }
          st[I_NUTT] += 1;
          st[I_DEST + k] = b.ed1; st[I_DELAY + k] = b.d1; conf[k] = b.dc1;
          st[I_DEST + k + 1] = b.ed2; st[I_DELAY + k + 1] = b.d2; conf[k + 1] = b.dc2;
          st[I_SSTART + k + 1] = st[I_SSTART + k]; st[I_SEND + k + 1] = st[I_SEND + k];
          if (b.d2 < b.d1) { st[I_START + k] = us; st[I_END + k] = b.bp; st[I_START + k + 1] = b.bp; st[I_END + k + 1] = ue; }
          else {
            const int half = cdiv(b.d2 - b.d1, 2 * ds);
            st[I_START + k] = us; st[I_END + k] = b.bp + half; st[I_START + k + 1] = b.bp - half; st[I_END + k + 1] = ue;
          }
          if ((st[I_START + k] - SEARCHBUFFER) * ds + b.d1 < 0) st[I_START + k] = SEARCHBUFFER + cdiv(ds - 1 - b.d1, ds);
          if (st[I_END + k + 1] * ds + b.d2 > S.nsamp[1] - SEARCHBUFFER * ds)
            st[I_END + k + 1] = cdiv(S.nsamp[1] - b.d2, ds) - SEARCHBUFFER;
        }
        T.sync();
      }
    }
    if (!split) ++k;
  }
}

// ---- perceptual model ----------------------------------------------------------------------------------------------------
// Bark power densities of one Hann-windowed frame of `data` -> ppd[0, nb).  la / lb: LDS transform buffers.
PQ_FN inline void pitch_pow_dens(const Team& T, const Params& P, const float* data, int start, bool valid, float* ppd, const Lds& L) {
  const int nf = 8 * P.ds, nb = P.nb;
  const Tables& tb = *P.tb;
  if (!valid) {
    for (int b = T.tid; b < nb; b += T.nt) ppd[b] = 0.f;
    T.sync();
    return;
  }
  for (int i = T.tid; i < nf; i += T.nt) {
    const float w = 0.5f * (1.0f - cosf(TWOPI_F * i / nf));
    float2 v;
    v.x = data[start + i] * w; v.y = 0.f;
    L.la[i] = v;
  }
  T.sync();
  float2* X = fft(T, L.la, L.lb, nf, false, P);
  for (int k = T.tid; k < nf / 2; k += T.nt) L.x[k] = k == 0 ? 0.f : X[k].x * X[k].x + X[k].y * X[k].y;
  T.sync();
  for (int b = T.tid; b < nb; b += T.nt) {
    double s = 0;
    const int k0 = tb.first_bin[b], n = tb.nr[b];
    for (int k = 0; k < n; ++k) s += L.x[k0 + k];
    ppd[b] = (float)(s * tb.pow_corr[b] * tb.sp);
  }
  T.sync();
}

PQ_FN inline float total_audible(const Params& P, const float* ppd, float factor) {       // one thread
  double r = 0;
  for (int b = 1; b < P.nb; ++b) {
    const float h = ppd[b];
    if (h > factor * P.tb->abs_thresh[b]) r += h;
  }
  return (float)r;
}

PQ_FN inline float loudness(const Params& P, int b, float input) {
  const Tables& tb = *P.tb;
  const float th = tb.abs_thresh[b];
  float h = tb.centre_bark[b] < 4.f ? 6.f / (tb.centre_bark[b] + 2.f) : 1.f;
  if (h > 2.f) h = 2.f;
  h = powf(h, 0.15f);
  const double zp = 0.23 * h;
  if (input > th) return (float)(pow(th / 0.5, zp) * (pow(0.5 + 0.5 * input / th, zp) - 1.0)) * tb.sl;
  return 0.f;
}

// symmetric / asymmetric frame disturbance of one frame (one thread; 49 bands)
PQ_FN inline void frame_disturbance(const Params& P, const float* pr, const float* pd, float* fd, float* fda) {
  const Tables& tb = *P.tb;
  double tw = 0, rs = 0, ra = 0;
  for (int b = 1; b < P.nb; ++b) {
    const float lr = loudness(P, b, pr[b]), ld = loudness(P, b, pd[b]);
    float d = ld - lr;
    const float m = 0.25f * fminf(ld, lr);
    if (d > m) d -= m; else if (d < -m) d += m; else d = 0.f;
    const float w = tb.width_bark[b];
    const float hs = fabsf(d) * w;
    rs += (double)hs * hs;
    const float ratio = (pd[b] + 50.f) / (pr[b] + 50.f);
    float h = powf(ratio, 1.2f);
    if (h > 12.f) h = 12.f;
    if (h < 3.f) h = 0.f;
    ra += fabsf(d * h) * w;
    tw += w;
  }
  *fd = (float)(sqrt(rs / tw) * tw);
  *fda = (float)(ra / tw * tw);
}

PQ_FN inline float lpq_weight(int start_frame, int stop_frame, float p_syl, float p_time, const float* fd, const float* twt) {
  double res = 0, tot = 0;
  for (int s = start_frame; s <= stop_frame; s += 10) {
    double rs = 0;
    for (int f = s; f < s + 20; ++f) if (f <= stop_frame) rs += pow((double)fd[f], (double)p_syl);
    rs = pow(rs / 20.0, 1.0 / p_syl);
    res += pow(twt[s - start_frame] * rs, (double)p_time);
    tot += pow((double)twt[s - start_frame], (double)p_time);
  }
  return (float)pow(res / tot, 1.0 / p_time);
}

PQ_FN inline int delay_at(const Params& P, const int* st, int sample) {
  int u = st[I_NUTT] - 1;
  while (u >= 0 && st[I_START + u] * P.ds > sample) --u;
  return u >= 0 ? st[I_DELAY + u] : st[I_DELAY];
}

// compute_delay of the bad-interval realignment: |x| cross-correlation through the long FFT
PQ_FN inline int compute_delay(const Team& T, const Params& P, Pair& S, const Lds& L, const float* s1, const float* s2, int n,
                               int search_range, float* max_corr) {
  int p2 = nextpow2(2 * n);
  if (p2 < 2048) p2 = 2048;
  const double pw1 = pow_of(T, s1, 0, n, n) * (double)n / p2, pw2 = pow_of(T, s2, 0, n, n) * (double)n / p2;
  if (pw1 <= 1e-6 || pw2 <= 1e-6 || p2 > S.p2max) { *max_corr = 0.f; return 0; }
  const double norm = sqrt(pw1 * pw2);
  for (int i = T.tid; i < p2; i += T.nt) {
    float2 v;
    v.x = i < n ? fabsf(s1[i]) : 0.f;
    v.y = i < n ? fabsf(s2[i]) : 0.f;
    S.ca[i] = v;
  }
  T.sync();
  const int n1 = p2 >> 10;
  const double pw_unused = 0; (void)pw_unused;
  fft_big_forward(T, P, S.ca, S.cb, p2, L.la, L.lb);
  float2* Z = S.cb;                                       // permuted order: Z[big_pos(k)]
  float2* O = S.ca;
  for (int k = T.tid; k < p2; k += T.nt) {
    const int pk = big_pos(k, n1);
    const float2 a = Z[pk], b = Z[big_pos((p2 - k) & (p2 - 1), n1)];
    const float x1r = 0.5f * (a.x + b.x) / p2, x1i = 0.5f * (a.y - b.y) / p2;
    const float x2r = 0.5f * (a.y + b.y), x2i = -0.5f * (a.x - b.x);
    float2 p;
    p.x = x1r * x2r + x1i * x2i;
    p.y = x1r * x2i - x1i * x2r;
    O[pk] = p;
  }
  T.sync();
  fft_big_inverse(T, P, O, Z, p2, L.la, L.lb);
  float2* R = Z;
  // candidates in the reference order: -search_range .. -1, then 0 .. search_range - 1; first maximum wins
  float lb = 0.f;
  int li = -1;
  for (int c = T.tid; c < 2 * search_range; c += T.nt) {
    const int i = c - search_range;
    const float h = (float)(fabsf(R[(i + p2) & (p2 - 1)].x / p2) / norm);
    if (h > lb) { lb = h; li = c; }
  }
  float best;
  int bidx;
  T.argmax(lb, li, &best, &bidx);
  if (bidx < 0 || best <= 0.f) { *max_corr = 0.f; return 0; }
  *max_corr = best;
  return bidx - search_range;
}

// the whole measurement of one pair; returns raw PESQ, fills the trace
PQ_FN inline float pesq_pair(const Team& T, const Params& P, Pair& S, const Lds& L, int* trace) {
  const int ds = P.ds, sb = SEARCHBUFFER * ds;
  int* st = S.st;
  const int maxn = S.nsamp[0] > S.nsamp[1] ? S.nsamp[0] : S.nsamp[1];
  for (int i = T.tid; i < I_COUNT; i += T.nt) st[i] = 0;
  for (int i = T.tid; i < F_COUNT; i += T.nt) S.fst[i] = 0.f;
  for (int i = T.tid; i < TRACE_INTS; i += T.nt) trace[i] = 0;
  T.sync();
  const long long t_begin = stamp();
  for (int sig = 0; sig < 2; ++sig) fix_power_level(T, P, S, L, sig, maxn);
  const long long t_level = stamp();
  // (both signals of a pair have the same length here: the batch API pads / crops them to a common `lens[pair]`)
  if (P.wb) iir_cascade2(T, S.data[0], S.data[1], S.nsamp[0] + P.pad, P.fs == 16000 ? WB_INIIR_16K : WB_INIIR_8K, 1, L.iir);
  else for (int sig = 0; sig < 2; ++sig) apply_filter(T, P, S, L, S.data[sig], S.nsamp[sig], STANDARD_IRS_FILTER_DB, 26);
  const long long t_input = stamp();
  for (int sig = 0; sig < 2; ++sig) {
    for (int i = T.tid; i < S.na; i += T.nt) S.adata[sig][i] = S.data[sig][i];
    T.sync();
    dc_block(T, P, S.adata[sig], S.nsamp[sig]);
  }
  if (P.fs == 16000) iir_cascade2(T, S.adata[0], S.adata[1], S.nsamp[0] + P.pad, INIIR_16K, 12, L.iir);
  else iir_cascade2(T, S.adata[0], S.adata[1], S.nsamp[0] + P.pad, INIIR_8K, 8, L.iir);
  const long long t_iir = stamp();
  for (int sig = 0; sig < 2; ++sig) apply_vad(T, P, S.adata[sig], S.nsamp[sig], S.vad[sig], S.logvad[sig], L);
  const long long t_filtered = stamp();
  crude_align(T, P, S, -1);
  if (T.tid == 0) st[I_NUTT] = utt_scan(P, S, true);
  T.sync();
  trace[0] = st[I_CRUDE];
  if (st[I_NUTT] < 1) { if (T.tid == 0) trace[1] = 0; T.sync(); return -1000.f; }
  const int nsearch = st[I_NUTT];
  for (int u = 0; u < nsearch; ++u) {
    crude_align(T, P, S, u);
    time_align(T, P, S, L, u);
  }
  const long long t_aligned = stamp();
  if (T.tid == 0) id_utterances(P, S);
  T.sync();
  utterance_split(T, P, S, L);
  T.sync();
  const long long t_split = stamp();
  const int nutt = st[I_NUTT];
  if (T.tid == 0) {
    trace[1] = nutt;
    for (int u = 0; u < nutt; ++u) { trace[8 + u] = st[I_START + u]; trace[8 + MAXNUTT + u] = st[I_END + u]; trace[8 + 2 * MAXNUTT + u] = st[I_DELAY + u]; }
  }
  // ---- perceptual model on the model signals ----
  const int nf = 8 * ds, half = nf / 2, nb = P.nb;
  const float* rd = S.data[0];
  const float* dd = S.data[1];
  if (T.tid == 0) {
    int skip_start = 0;
    float s5;
    do {
      s5 = 0.f;
      for (int i = 0; i < 5; ++i) s5 += fabsf(rd[sb + skip_start + i]);
      if (s5 < 500.f) ++skip_start;
    } while (s5 < 500.f && skip_start < maxn / 2);
    int skip_end = 0;
    do {
      s5 = 0.f;
      for (int i = 0; i < 5; ++i) s5 += fabsf(rd[maxn - sb + P.pad - 1 - skip_end - i]);
      if (s5 < 500.f) ++skip_end;
    } while (s5 < 500.f && skip_end < maxn / 2);
    st[I_TMP] = skip_start / half;
    st[I_TMP + 1] = (maxn - 2 * sb + P.pad) / half - 1 - skip_end / half;
  }
  T.sync();
  const int start_frame = st[I_TMP], stop_frame = st[I_TMP + 1];
  const int total_frames = (maxn - 2 * sb + P.pad) / half - 1;
  const int nfr = stop_frame + 1;
  int* silent = reinterpret_cast<int*>(S.scratch);                 // nfr ints
  float* twt = S.scratch + nfr;                                    // nfr floats
  float* avg = twt + nfr;                                          // 3 * nb floats
  for (int f = 0; f < nfr; ++f) {
    const int s_ref = sb + f * half;
    pitch_pow_dens(T, P, rd, s_ref, true, S.ppd_ref + (long)f * nb, L);
    const int s_deg = s_ref + delay_at(P, st, s_ref);
    pitch_pow_dens(T, P, dd, s_deg, s_deg > 0 && s_deg + nf < maxn + P.pad, S.ppd_deg + (long)f * nb, L);
  }
  for (int f = T.tid; f < nfr; f += T.nt) silent[f] = total_audible(P, S.ppd_ref + (long)f * nb, 1e2f) < 1e7f;
  T.sync();
  for (int b = T.tid; b < nb; b += T.nt) {
    double ar = 0, ad = 0;
    const float th = 100.f * P.tb->abs_thresh[b];
    for (int f = 0; f < nfr; ++f) {
      if (silent[f]) continue;
      const float hr = S.ppd_ref[(long)f * nb + b], hd = S.ppd_deg[(long)f * nb + b];
      if (hr > th) ar += hr;
      if (hd > th) ad += hd;
    }
    float x = ((float)(ad / total_frames) + 1000.f) / ((float)(ar / total_frames) + 1000.f);
    if (x > 100.f) x = 100.f;
    if (x < 0.01f) x = 0.01f;
    avg[b] = x;
  }
  T.sync();
  for (long i = T.tid; i < (long)nfr * nb; i += T.nt) S.ppd_ref[i] *= avg[i % nb];
  T.sync();
  // gain recursion over the frames (thread 0), then the disturbances in parallel over the frames
  float* ta_r = avg + 3 * nb;                                      // nfr floats each: audible powers of the frames
  float* ta_d = ta_r + nfr;
  auto scale_pass = [&](int f0, int f1, bool first) {
    for (int f = f0 + T.tid; f < f1; f += T.nt) {
      ta_r[f] = total_audible(P, S.ppd_ref + (long)f * nb, 1.f);
      ta_d[f] = total_audible(P, S.ppd_deg + (long)f * nb, 1.f);
    }
    T.sync();
    if (T.tid == 0) {
      float old = 1.f;
      for (int f = f0; f < f1; ++f) {
        const float ta_ref = ta_r[f], ta_deg = ta_d[f];
        if (first) S.tpr[f] = ta_ref;
        float sc = (ta_ref + 5e3f) / (ta_deg + 5e3f);
        if (f > 0) sc = 0.2f * old + 0.8f * sc;
        old = sc;
        if (sc > 5.0f) sc = 5.0f;
        if (sc < 3e-4f) sc = 3e-4f;
        twt[f] = sc;                     // (twt doubles as the per-frame scale until the time weights are written)
      }
    }
    T.sync();
    for (long i = (long)f0 * nb + T.tid; i < (long)f1 * nb; i += T.nt) S.ppd_deg[i] *= twt[i / nb];
    T.sync();
    for (int f = f0 + T.tid; f < f1; f += T.nt) {
      float a, b;
      frame_disturbance(P, S.ppd_ref + (long)f * nb, S.ppd_deg + (long)f * nb, &a, &b);
      if (first) { S.fd[f] = a; S.fda[f] = b; }
      else { S.fd[f] = fminf(S.fd[f], a); S.fda[f] = fminf(S.fda[f], b); }
    }
    T.sync();
  };
  scale_pass(0, nfr, true);
  float mxd = 0.f;
  for (int f = T.tid; f < nfr; f += T.nt) mxd = fmaxf(mxd, S.fd[f]);
  const bool bad_frame = T.maxf(mxd) > 30.f;
  if (T.tid == 0) {
    for (int u = 1; u < nutt; ++u) {
      int frame1 = (int)floor((double)((st[I_START + u] - SEARCHBUFFER) * ds + st[I_DELAY + u]) / half);
      const int j = (int)floor((double)((st[I_END + u - 1] - SEARCHBUFFER) * ds + st[I_DELAY + u - 1])) / half;
      const int jump = st[I_DELAY + u] - st[I_DELAY + u - 1];
      if (frame1 > j) frame1 = j;
      if (frame1 < 0) frame1 = 0;
      if (jump < -half) {
        const int aj = jump < 0 ? -jump : jump;
        const int frame2 = ((st[I_START + u] - SEARCHBUFFER) * ds + aj) / half + 1;
        for (int f = frame1; f <= frame2; ++f) if (f < stop_frame) { S.fd[f] = 0.f; S.fda[f] = 0.f; }
      }
    }
  }
  T.sync();
  int nbad = 0;
  if (bad_frame) {
    const int nn = P.pad + maxn;
    for (int i = T.tid; i < nn; i += T.nt) {
      float v = 0.f;
      if (i >= sb && i < nn - sb) {
        int j = i + delay_at(P, st, i);
        if (j < sb) j = sb;
        if (j >= nn - sb) j = nn - sb - 1;
        v = dd[j];
      }
      S.tweaked[i] = v;
    }
    T.sync();
    int* bi = st + I_TMP + 2;                                    // [count, (start, stop) x MAXBAD/...]: kept in trace instead
    if (T.tid == 0) {
      // is_bad / smeared in the silent[] array (ints): bit 0 = bad, bit 1 = smeared
      for (int f = 0; f < nfr; ++f) silent[f] = S.fd[f] > 30.f ? 1 : 0;
      silent[0] = 0;
      for (int f = 2; f < stop_frame - 2; ++f) {
        const int l = (silent[f - 2] | silent[f - 1] | silent[f]) & 1, r = (silent[f] | silent[f + 1] | silent[f + 2]) & 1;
        if (l && r) silent[f] |= 2;
      }
      int f = 0, cnt = 0;
      while (f <= stop_frame) {
        while (f <= stop_frame && !(silent[f] & 2)) ++f;
        if (f <= stop_frame) {
          const int s0 = f;
          while (f <= stop_frame && (silent[f] & 2)) ++f;
          if (f <= stop_frame && f - s0 >= 5 && cnt < MAXBAD) { trace[8 + 3 * MAXNUTT + 2 * cnt] = s0; trace[8 + 3 * MAXNUTT + 2 * cnt + 1] = f; ++cnt; }
        }
      }
      bi[0] = cnt;
    }
    T.sync();
    nbad = bi[0];
    const int srange = 4 * nf;
    float* r = S.doubly;                      // staging for the two series of compute_delay (free until the copy below)
    for (int q = 0; q < nbad; ++q) {
      const int f0 = trace[8 + 3 * MAXNUTT + 2 * q], f1 = trace[8 + 3 * MAXNUTT + 2 * q + 1];
      const int st_s = f0 * half + sb, sp_s = f1 * half + nf + sb, ns = sp_s - st_s, len = 2 * srange + ns;
      int dly = 0;
      if (2 * len <= S.na) {
        float* d = r + len;
        for (int i = T.tid; i < len; i += T.nt) {
          r[i] = (i >= srange && i < srange + ns) ? rd[st_s + i - srange] : 0.f;
          int j = st_s - srange + i;
          const int lim = maxn - sb + P.pad;
          if (j < sb) j = sb;
          if (j >= lim) j = lim - 1;
          d[i] = S.tweaked[j];
        }
        T.sync();
        float corr;
        dly = compute_delay(T, P, S, L, r, d, len, srange, &corr);
        if (corr < 0.5f) dly = 0;
      }
      if (T.tid == 0) bi[1 + q] = dly;
      T.sync();
    }
    if (nbad > 0) {
      for (int i = T.tid; i < maxn + P.pad; i += T.nt) S.doubly[i] = S.tweaked[i];
      T.sync();
      for (int q = 0; q < nbad; ++q) {
        const int f0 = trace[8 + 3 * MAXNUTT + 2 * q], f1 = trace[8 + 3 * MAXNUTT + 2 * q + 1];
        const int st_s = f0 * half + sb, sp_s = f1 * half + nf + sb, dly = bi[1 + q];
        for (int i = st_s + T.tid; i < sp_s; i += T.nt) {
          int j = i + dly;
          if (j < 0) j = 0;
          if (j >= maxn) j = maxn - 1;
          S.doubly[i] = S.tweaked[j];
        }
        T.sync();
      }
      for (int q = 0; q < nbad; ++q) {
        const int f0 = trace[8 + 3 * MAXNUTT + 2 * q], f1 = trace[8 + 3 * MAXNUTT + 2 * q + 1];
        for (int f = f0; f < f1; ++f) pitch_pow_dens(T, P, S.doubly, sb + f * half, true, S.ppd_deg + (long)f * nb, L);
        scale_pass(f0, f1, false);
      }
    }
  }
  for (int f = T.tid; f < nfr; f += T.nt) {
    float h = 1.f;
    if (nfr > 1000) {
      const int n = (maxn - 2 * sb) / half - 1;
      float twf = (n - 1000.f) / 5500.f;
      if (twf > 0.5f) twf = 0.5f;
      h = (1.0f - twf) + twf * (float)f / (float)n;
    }
    twt[f] = h;
    const float g = (float)pow((S.tpr[f] + 1e5) / 1e7, 0.04);
    float a = S.fd[f] / g, b = S.fda[f] / g;
    if (a > 45.f) a = 45.f;
    if (b > 45.f) b = 45.f;
    S.fd[f] = a; S.fda[f] = b;
  }
  T.sync();
  float raw = 0.f;
  if (T.tid == 0) {
    const float d_ind = lpq_weight(start_frame, stop_frame, 6.f, 2.f, S.fd, twt);
    const float a_ind = lpq_weight(start_frame, stop_frame, 6.f, 2.f, S.fda, twt);
    raw = 4.5f - 0.1f * d_ind - 0.0309f * a_ind;
    S.fst[F_TMP] = raw;
    trace[2] = start_frame; trace[3] = stop_frame; trace[4] = nbad;
    // stage timers, 16 bits each in units of 64 us (100 MHz wall clock): trace[5] = level filters | input filter,
    // trace[6] = DC + alignment IIR | VAD, trace[7] = crude / fine alignment + utterance splitting | perceptual model
    const long long t_end = stamp();
    auto u = [](long long dt) { return (int)(dt / 6400) & 0xffff; };
    trace[5] = u(t_level - t_begin) | (u(t_input - t_level) << 16);
    trace[6] = u(t_iir - t_input) | (u(t_filtered - t_iir) << 16);
    trace[7] = u(t_split - t_filtered) | (u(t_end - t_split) << 16);
    (void)t_aligned;
  }
  T.sync();
  return S.fst[F_TMP];
}

}  // namespace pesq
