// LDS-resident mixed-radix Stockham FFT used by the framed STFT / iSTFT / loss kernels.
//
// A workgroup transforms `nf` independent complex sequences of length n that sit in LDS
// (float2 a[nf*n]), ping-ponging with a second buffer.  n = prod(radix[]), radices in
// {2,3,4,5} are closed-form butterflies, anything else (7, 49 ... for n_fft=441/882) goes
// through the generic O(R^2) butterfly.  Twiddles W_n^j (j<n) are read from an LDS copy.
// Two real frames are packed per complex sequence by the callers (real -> re, next frame -> im),
// so one complex FFT yields two one-sided spectra.
#pragma once
#include "urse_common.h"

namespace urse {

struct FftPlan {
  int n;
  int nrad;
  int radix[12];
  // magic multipliers: floor(x / d) == __umulhi(x, m) for x, d < 2^16 (m = floor(2^32 / d) + 1); integer division
  // has no hardware instruction on CDNA and dominated the index arithmetic of these kernels
  unsigned m_n, m_f;            // d = n, d = n/2 + 1
  unsigned m_nb[12], m_ns[12];  // d = n / radix[s], d = product of the previous radices
};

static inline unsigned fastdiv_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d) + 1u; }
__device__ __forceinline__ int fastdiv(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

static inline bool make_fft_plan(int n, FftPlan* p) {
  p->n = n;
  p->nrad = 0;
  int m = n;
  while (m % 4 == 0) { p->radix[p->nrad++] = 4; m /= 4; }
  while (m % 2 == 0) { p->radix[p->nrad++] = 2; m /= 2; }
  while (m % 3 == 0) { p->radix[p->nrad++] = 3; m /= 3; }
  while (m % 5 == 0) { p->radix[p->nrad++] = 5; m /= 5; }
  for (int f = 7; f <= 61 && m > 1; f += 2)
    while (m % f == 0) { if (p->nrad >= 12) return false; p->radix[p->nrad++] = f; m /= f; }
  if (!(m == 1 && p->nrad <= 12) || n >= 65536) return false;
  p->m_n = fastdiv_magic((unsigned)n);
  p->m_f = fastdiv_magic((unsigned)(n / 2 + 1));
  int Ns = 1;
  for (int s = 0; s < p->nrad; ++s) {
    p->m_nb[s] = fastdiv_magic((unsigned)(n / p->radix[s]));
    p->m_ns[s] = fastdiv_magic((unsigned)Ns);
    Ns *= p->radix[s];
  }
  return true;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// Forward DFT (kernel e^{-2 pi i jk/n}).  Returns the buffer that holds the result.
// All threads of the block must call; ends with a __syncthreads().
__device__ inline float2* fft_lds_forward(float2* a, float2* b, int nf, const FftPlan& plan,
                                          const float2* __restrict__ tw /* LDS, n entries */) {
  const int n = plan.n;
  int Ns = 1;
  for (int s = 0; s < plan.nrad; ++s) {
    const int R = plan.radix[s];
    const int nb = n / R;
    const int total = nf * nb;
    const int tstride = n / (Ns * R);
    const unsigned mnb = plan.m_nb[s], mns = plan.m_ns[s];
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
      const int f = fastdiv(idx, mnb);
      const int j = idx - f * nb;
      const int k = j - fastdiv(j, mns) * Ns;
      const float2* in = a + f * n + j;
      float2* out = b + f * n + (j - k) * R + k;
      const int tk = k * tstride;
      if (R == 4) {
        float2 v0 = in[0];
        float2 v1 = cmul(in[nb], tw[tk]);
        float2 v2 = cmul(in[2 * nb], tw[2 * tk]);
        float2 v3 = cmul(in[3 * nb], tw[3 * tk]);
        float2 s02 = cadd(v0, v2), d02 = csub(v0, v2);
        float2 s13 = cadd(v1, v3), d13 = cmul_mi(csub(v1, v3));
        out[0] = cadd(s02, s13);
        out[Ns] = cadd(d02, d13);
        out[2 * Ns] = csub(s02, s13);
        out[3 * Ns] = csub(d02, d13);
      } else if (R == 2) {
        float2 v0 = in[0];
        float2 v1 = cmul(in[nb], tw[tk]);
        out[0] = cadd(v0, v1);
        out[Ns] = csub(v0, v1);
      } else if (R == 3) {
        float2 v0 = in[0];
        float2 v1 = cmul(in[nb], tw[tk]);
        float2 v2 = cmul(in[2 * nb], tw[2 * tk]);
        float2 s = cadd(v1, v2);
        float2 d = csub(v1, v2);
        const float c = -0.5f, sn = 0.86602540378443864676f;
        float2 m = make_float2(v0.x + c * s.x, v0.y + c * s.y);
        float2 r = make_float2(sn * d.y, -sn * d.x);  // -i*sin(60)*d
        out[0] = cadd(v0, s);
        out[Ns] = cadd(m, r);
        out[2 * Ns] = csub(m, r);
      } else if (R == 5) {
        float2 v0 = in[0];
        float2 v1 = cmul(in[nb], tw[tk]);
        float2 v2 = cmul(in[2 * nb], tw[2 * tk]);
        float2 v3 = cmul(in[3 * nb], tw[3 * tk]);
        float2 v4 = cmul(in[4 * nb], tw[4 * tk]);
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        float2 a14 = cadd(v1, v4), b14 = csub(v1, v4);
        float2 a23 = cadd(v2, v3), b23 = csub(v2, v3);
        out[0] = make_float2(v0.x + a14.x + a23.x, v0.y + a14.y + a23.y);
        float2 p1 = make_float2(v0.x + c1 * a14.x + c2 * a23.x, v0.y + c1 * a14.y + c2 * a23.y);
        float2 p2 = make_float2(v0.x + c2 * a14.x + c1 * a23.x, v0.y + c2 * a14.y + c1 * a23.y);
        // q = -i*(s1*b14 + s2*b23),  q' = -i*(s2*b14 - s1*b23)
        float2 q1 = make_float2(s1 * b14.y + s2 * b23.y, -(s1 * b14.x + s2 * b23.x));
        float2 q2 = make_float2(s2 * b14.y - s1 * b23.y, -(s2 * b14.x - s1 * b23.x));
        out[Ns] = cadd(p1, q1);
        out[4 * Ns] = csub(p1, q1);
        out[2 * Ns] = cadd(p2, q2);
        out[3 * Ns] = csub(p2, q2);
      } else {
        const int rs = n / R;
        for (int q = 0; q < R; ++q) {
          float2 acc = in[0];
          int qr = 0;
          for (int r = 1; r < R; ++r) {
            qr += q;
            if (qr >= R) qr -= R;
            acc = cadd(acc, cmul(cmul(in[r * nb], tw[r * tk]), tw[qr * rs]));
          }
          out[q * Ns] = acc;
        }
      }
    }
    __syncthreads();
    float2* t = a; a = b; b = t;
    Ns *= R;
  }
  return a;
}

}  // namespace urse
