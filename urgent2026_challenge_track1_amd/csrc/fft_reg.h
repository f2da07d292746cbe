// Register-resident small DFTs for the two-pass "32 x M" FFTs (one half-wave per transform): a lane holds all points of one
// short DFT in registers, every index below is a compile-time constant after unrolling.  Forward kernel e^{-2 pi i jk/P}.
// (stft.hip keeps its own 32- and 30-point forms for the 960-point STFT; these serve the MR-L1 loss windows 256 / 512 / 768 / 1024.)
#pragma once
#include "urse_common.h"

namespace urse {

constexpr __device__ float FR_C32[16] = {1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f,
                                         0.382683432f, 0.195090322f, 0.000000000f, -0.195090322f, -0.382683432f, -0.555570233f,
                                         -0.707106781f, -0.831469612f, -0.923879533f, -0.980785280f};
constexpr __device__ float FR_S32[16] = {0.000000000f, 0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f,
                                         0.923879533f, 0.980785280f, 1.000000000f, 0.980785280f, 0.923879533f, 0.831469612f,
                                         0.707106781f, 0.555570233f, 0.382683432f, 0.195090322f};

constexpr __host__ __device__ int fr_log2(int p) { return p <= 1 ? 0 : 1 + fr_log2(p >> 1); }
constexpr __host__ __device__ int fr_brev(int k, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((k >> i) & 1) << (bits - 1 - i);
  return r;
}

// P-point DFT (P = 2, 4, 8, 16, 32), radix-2 decimation in frequency, in place: natural order in, BIT-REVERSED order out:
// X[k] = v[fr_brev(k, log2 P)]
template <int P>
__device__ __forceinline__ void dft_pow2_dif(float2 (&v)[P]) {
  constexpr int LG = fr_log2(P), TS = 32 / P;        // W_P^m = W_32^(m * TS)
#pragma unroll
  for (int s = 0; s < LG; ++s) {
    const int half = (P / 2) >> s;
#pragma unroll
    for (int blk = 0; blk < (1 << s); ++blk)
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const int i0 = blk * 2 * half + j, i1 = i0 + half;
        const float2 a = v[i0], b = v[i1];
        v[i0] = make_float2(a.x + b.x, a.y + b.y);
        const float dx = a.x - b.x, dy = a.y - b.y;
        const int m = (j << s) * TS;                   // twiddle W_32^m = cos - i sin, m in [0, 16)
        if (m == 0) v[i1] = make_float2(dx, dy);
        else if (m == 8) v[i1] = make_float2(dy, -dx);
        else v[i1] = make_float2(dx * FR_C32[m] + dy * FR_S32[m], dy * FR_C32[m] - dx * FR_S32[m]);
      }
  }
}

__device__ __forceinline__ void fr_dft3(float2& v0, float2& v1, float2& v2) {
  const float sn = 0.86602540378443864676f;
  const float2 sm = make_float2(v1.x + v2.x, v1.y + v2.y), d = make_float2(v1.x - v2.x, v1.y - v2.y);
  const float2 m = make_float2(v0.x - 0.5f * sm.x, v0.y - 0.5f * sm.y);
  const float2 r = make_float2(sn * d.y, -sn * d.x);
  v0 = make_float2(v0.x + sm.x, v0.y + sm.y);
  v1 = make_float2(m.x + r.x, m.y + r.y);
  v2 = make_float2(m.x - r.x, m.y - r.y);
}

// M-point DFT of u[0 .. M), natural order in, natural order out (M = 8, 16, 32: radix-2; M = 24: 3 x 8 prime-factor transform,
// input map n = (8 n1 + 3 n2) mod 24, output map k = (16 k1 + 9 k2) mod 24 - no twiddles between the factors)
template <int M>
__device__ __forceinline__ void dft_small(float2 (&u)[M]) {
  if constexpr (M == 24) {
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) fr_dft3(u[(3 * n2) % 24], u[(8 + 3 * n2) % 24], u[(16 + 3 * n2) % 24]);
    float2 o[24];
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
      float2 w[8];
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) w[n2] = u[(8 * k1 + 3 * n2) % 24];
      dft_pow2_dif<8>(w);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) o[(16 * k1 + 9 * k2) % 24] = w[fr_brev(k2, 3)];
    }
#pragma unroll
    for (int k = 0; k < 24; ++k) u[k] = o[k];
  } else {
    dft_pow2_dif<M>(u);
    float2 o[M];
#pragma unroll
    for (int k = 0; k < M; ++k) o[k] = u[fr_brev(k, fr_log2(M))];
#pragma unroll
    for (int k = 0; k < M; ++k) u[k] = o[k];
  }
}

}  // namespace urse
