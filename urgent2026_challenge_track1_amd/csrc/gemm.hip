// Dense contractions of the BSRNN hot path on the CDNA4 matrix cores.
//
//   gemm_nt : C[M,N]  = epi( A[M,K] * B[N,K]^T )         forward projections, dgrads (weights pre-transposed)
//   gemm_tn : C[Mo,No] += A[R,Mo]^T * B[R,No]            weight gradients (reduction over the row axis, split-R)
//
// Operands are bf16 (v_mfma_f32_16x16x32_bf16) or f32 (v_mfma_f32_16x16x4_f32, exact f32); accumulation
// is always f32.  128x128 workgroup tiles, 4 waves of 64x64 (4x4 MFMA tiles), K staged through
// double-buffered LDS in 64-byte slabs with register prefetch (one barrier per K-step).  Grouped
// launches (one descriptor per band) serve the band-split / mask-decoder 1x1 convolutions.
// They replace the cuBLAS calls under nn.Linear / nn.Conv1d(k=1) / nn.LSTM input projections of
// espnet2's BSRNN (reference twin: baseline_code/models/bsrnn_flowse.py:66-81,296-307).
#include <stdlib.h>

#include <type_traits>
#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct GemmDesc {  // 12 x int64, mirrored by ops.py
  const char* A;
  const char* B;
  char* C;
  const float* bias;
  const float* resid;
  long lda, ldb, ldc;  // elements
  long M, N, K;
  long ldr;            // resid leading dim (elements)
};

constexpr int BM = 128, BN = 128;
constexpr int SLAB = 64;      // bytes of K per LDS row

// bijective XCD-aware remap: blocks that share an XCD (bid % 8) get consecutive tile ids
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef short8_t type;
  static constexpr int KSUB = 1;
  static __device__ __forceinline__ type load(const char* row, int lane, int) {
    return *reinterpret_cast<const short8_t*>(row + 16 * (lane >> 4));
  }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
  }
};
template <> struct Frag<f16_t> {        // IEEE half operands (forward kernels of the f16 mode): same fragments, v_mfma_f32_16x16x32_f16
  typedef short8_t type;
  static constexpr int KSUB = 1;
  static __device__ __forceinline__ type load(const char* row, int lane, int) {
    return *reinterpret_cast<const short8_t*>(row + 16 * (lane >> 4));
  }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Frag<float> {
  typedef float type;
  static constexpr int KSUB = 4;
  static __device__ __forceinline__ type load(const char* row, int lane, int s) {
    return *reinterpret_cast<const float*>(row + 16 * s + 4 * (lane >> 4));
  }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

template <typename T, typename TO>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const GemmDesc* __restrict__ descs, GemmDesc single, int act) {
  // K is staged in 128-byte slabs (two 64-byte MFMA sub-slabs) per barrier; rows padded to 144 B
  constexpr int SLAB2 = 2 * SLAB, PITCH = SLAB2 + 16;
  __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * PITCH];
  const GemmDesc d = descs ? descs[blockIdx.z] : single;
  const int tn = (int)((d.N + BN - 1) / BN), tm = (int)((d.M + BM - 1) / BM);
  const int nblk = tm * tn;
  if ((int)blockIdx.x >= nblk) return;
  const int bid = xcd_remap(blockIdx.x, nblk);
  const int tile_m = bid / tn, tile_n = bid - tile_m * tn;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  constexpr int ES = sizeof(T);
  const long Kb = d.K * ES;  // bytes of K (multiple of 64)
  const int nk = (int)((Kb + SLAB2 - 1) / SLAB2);

  auto As = [&](int buf) -> char* { return lds + buf * (BM + BN) * PITCH; };
  auto Bs = [&](int buf) -> char* { return lds + buf * (BM + BN) * PITCH + BM * PITCH; };

  // staging map: 4 chunks (16 B) of A and 4 of B per thread; a row's 128 B are read by 8 consecutive lanes
  const int srow = tid >> 3, skc = tid & 7;
  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const long kb = (long)kt * SLAB2 + skc * 16;
    const bool kin = kb < Kb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = m0 + srow + 32 * i;
      ra[i] = (kin && row < d.M) ? *reinterpret_cast<const uint4*>(d.A + (row * d.lda) * ES + kb) : make_uint4(0, 0, 0, 0);
      const long col = n0 + srow + 32 * i;
      rb[i] = (kin && col < d.N) ? *reinterpret_cast<const uint4*>(d.B + (col * d.ldb) * ES + kb) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(As(buf) + (srow + 32 * i) * PITCH + skc * 16) = ra[i];
      *reinterpret_cast<uint4*>(Bs(buf) + (srow + 32 * i) * PITCH + skc * 16) = rb[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* a_base = As(cur) + (wm * 64 + (lane & 15)) * PITCH;
    const char* b_base = Bs(cur) + (wn * 64 + (lane & 15)) * PITCH;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = 0; s < Frag<T>::KSUB; ++s) {
        typename Frag<T>::type a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = Frag<T>::load(a_base + i * 16 * PITCH + h * SLAB, lane, s);
          b[i] = Frag<T>::load(b_base + i * 16 * PITCH + h * SLAB, lane, s);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Frag<T>::mma(a[i], b[j], acc[i][j]);
      }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: stage the tile through LDS so that global stores (and the residual read) are full-row,
  // 16-byte-per-lane coalesced instead of 2/4-byte scattered in the MFMA C layout ----
  constexpr int OS = sizeof(TO);
  constexpr int CP = BN * OS + 16;                 // staged row pitch
  constexpr int RPP = (OS == 2) ? 128 : 64;        // rows per pass (fits the 72 KiB staging buffer)
  constexpr int EPC = 16 / OS;                     // elements per 16-byte chunk
  constexpr int CPR = BN / EPC;                    // chunks per row
  TO* C = reinterpret_cast<TO*>(d.C);
  // f16 forward mode, act 1 (the mask decoder's tanh layer): the aux slot (resid, ldr) names a SECOND output, the same values in bf16 -
  // the backward (tanh derivative, weight-gradient GEMM against bf16 gradients) reads that copy
  const bool c2 = __is_same(T, f16_t) && OS == 2 && act == 1 && d.resid != nullptr;
  bf16_t* C2 = c2 ? reinterpret_cast<bf16_t*>(const_cast<float*>(d.resid)) : nullptr;
  const bool vec_ok = ((d.ldc * OS) % 16 == 0) && ((reinterpret_cast<uintptr_t>(d.C) % 16) == 0) &&
                      (!d.resid || act == 2 || (((d.ldr * 4) % 16 == 0) && (reinterpret_cast<uintptr_t>(d.resid) % 16) == 0)) &&
                      (!c2 || (d.ldr * 2) % 16 == 0);
#pragma unroll
  for (int pass = 0; pass < BM / RPP; ++pass) {
    if (pass > 0) __syncthreads();
    if (OS == 2 || wm == pass) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lcol = wn * 64 + j * 16 + (lane & 15);
        const long col = n0 + lcol;
        const float bv = (d.bias && col < d.N) ? d.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int lrow = (OS == 2 ? wm * 64 : 0) + i * 16 + (lane >> 4) * 4 + r;
            float v = acc[i][j][r] + bv;
            if (act == 1) v = tanhf_(v);
            if (act == 2) {  // tanh backward: aux (= resid slot, TO typed) holds h = tanh(.)
              const long row = m0 + (OS == 2 ? 0 : pass * RPP) + lrow;
              if (row < d.M && col < d.N) {
                const float hv = to_f32<TO>(reinterpret_cast<const TO*>(d.resid)[row * d.ldr + col]);
                v *= (1.f - hv * hv);
              }
            }
            *reinterpret_cast<TO*>(lds + lrow * CP + lcol * OS) = from_f32<TO>(v);
          }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < RPP * CPR; idx += 256) {
      const int lrow = idx / CPR, ch = idx - lrow * CPR;
      const long row = m0 + pass * RPP + lrow, col = n0 + ch * EPC;
      if (row >= d.M || col >= d.N) continue;
      const char* src = lds + lrow * CP + ch * 16;
      if (vec_ok && col + EPC <= d.N) {
        uint4 v = *reinterpret_cast<const uint4*>(src);
        if (OS == 4 && d.resid && act != 2) {
          const float4 rr = *reinterpret_cast<const float4*>(d.resid + row * d.ldr + col);
          float4 f = *reinterpret_cast<float4*>(&v);
          f.x += rr.x; f.y += rr.y; f.z += rr.z; f.w += rr.w;
          v = *reinterpret_cast<uint4*>(&f);
        }
        *reinterpret_cast<uint4*>(C + row * d.ldc + col) = v;
        if constexpr (__is_same(T, f16_t) && OS == 2) {
          if (c2) {
            float a0, a1, a2, a3, a4, a5, a6, a7;
            unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
            *reinterpret_cast<uint4*>(C2 + row * d.ldr + col) =
                make_uint4(pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7));
          }
        }
      } else {
        const TO* sv = reinterpret_cast<const TO*>(src);
        for (int e = 0; e < EPC && col + e < d.N; ++e) {
          float f = to_f32<TO>(sv[e]);
          if (OS == 4 && d.resid && act != 2) f += d.resid[row * d.ldr + col + e];
          C[row * d.ldc + col + e] = from_f32<TO>(f);
          if (c2) C2[row * d.ldr + col + e] = f32_to_bf16(f);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TN: C[Mo,No] += sum_r A[r,Mo] * Bsh[r,No],  Bsh[r] = valid(r) ? B[r+shift] : 0
// valid(r): period == 0, or ((r / inner) % period) != invalid_step  (h_{t-1} of the first step is 0)
constexpr int TROW_BF16 = 272;  // LDS row pitch for a [32 r][128 m] bf16 slab (256 + 16)
constexpr int TROW_F32 = 528;   // [16 r][128 m] f32 slab (512 + 16)

struct TnArgs {
  const char* A; const char* B; float* C; float* colsum;
  long lda, ldb, ldc;
  long R, Mo, No;
  long shift, inner, period, invalid_step;
  long rows_per_slice;
  long perm_h;   // > 0: A columns are gate-interleaved (dir, unit, gate); C rows / colsum are written as (dir, gate, unit)
  // dual-operand mode of the big kernel (B2 != nullptr): the first nt1 column tiles multiply A^T with B (plain rows,
  // colsum) into C, the others with B2 (row shift / step mask) into C2 -- one pass over A (the [M, 4H] dgates of one
  // direction) yields both dW_ih and dW_hh
  const char* B2; float* C2;
  long ldb2, ldc2, No2, nt1;
  long pad_[2];
};

__device__ __forceinline__ long tn_perm(long m, long h) {
  if (h == 0) return m;
  if (h < 0) {           // flow grad decoder: A columns ordered (bin, 16 sub-channels) -> weight rows (sub-channel, bin)
    const long sb = -h;
    return (m & 15) * sb + (m >> 4);
  }
  const long g4 = 4 * h, d = m / g4, r = m - d * g4;
  return d * g4 + (r & 3) * h + (r >> 2);
}

template <typename T>
__device__ __forceinline__ void gemm_tn_body(const TnArgs& p, const int bx, const int nblk, char* lds) {
  constexpr int ES = sizeof(T);
  constexpr int BKR = (ES == 2) ? 32 : 16;  // rows per step
  constexpr int PITCH = (ES == 2) ? TROW_BF16 : TROW_F32;
  const int tn = (int)((p.No + BN - 1) / BN);
  // XCD-aware order: the workgroups that land on one XCD (blockIdx % 8) take consecutive (slice, tile) ids, so all
  // output tiles of one row slice stream the same A / B rows through the same L2 at the same time (the operands are
  // re-read once per tile column / row; spread over the eight L2s those re-reads were served at Infinity-Cache rate)
  const int tiles = tn * (int)((p.Mo + BM - 1) / BM);
  const int lid = xcd_remap(bx, nblk);
  const int slice = lid / tiles, tile = lid - slice * tiles;
  const int tile_m = tile / tn, tile_n = tile - tile_m * tn;
  const long m0 = (long)tile_m * BM, n0 = (long)tile_n * BN;
  const long r_begin = (long)slice * p.rows_per_slice;
  long r_end = r_begin + p.rows_per_slice;
  if (r_end > p.R) r_end = p.R;
  if (r_begin >= r_end) return;
  const int nk = (int)((r_end - r_begin + BKR - 1) / BKR);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;

  auto As = [&](int buf) -> char* { return lds + buf * 2 * BKR * PITCH; };
  auto Bs = [&](int buf) -> char* { return lds + buf * 2 * BKR * PITCH + BKR * PITCH; };

  constexpr int CPR = BM * ES / 16;           // 16-B chunks per slab row (16 bf16 / 32 f32)
  constexpr int NCH = BKR * CPR / 256;        // chunks per thread per operand (2 / 2)
  uint4 ra[NCH], rb[NCH];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i;
      const int row = c / CPR, mc = c - row * CPR;
      const long r = r_begin + (long)kt * BKR + row;
      const long ca = m0 + mc * (16 / ES), cb = n0 + mc * (16 / ES);
      // guards on Mo / No, not on the pitch: A and B may be COLUMN SLICES of wider matrices (per-band operands), where "column < lda"
      // runs past the parent's row end - and, on the last row, past the allocation (a chunk that starts below Mo / No stays inside the
      // parent row: operands and pitches are 16-byte aligned)
      ra[i] = (r < r_end && ca < p.Mo) ? *reinterpret_cast<const uint4*>(p.A + (r * p.lda + ca) * ES)
                                        : make_uint4(0, 0, 0, 0);
      bool ok = (r < r_end && cb < p.No);
      const long rs = r + p.shift;
      if (p.period) ok = ok && (((r / p.inner) % p.period) != p.invalid_step);
      ok = ok && rs >= 0 && rs < p.R;
      rb[i] = ok ? *reinterpret_cast<const uint4*>(p.B + (rs * p.ldb + cb) * ES) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i;
      const int row = c / CPR, mc = c - row * CPR;
      *reinterpret_cast<uint4*>(As(buf) + row * PITCH + mc * 16) = ra[i];
      *reinterpret_cast<uint4*>(Bs(buf) + row * PITCH + mc * 16) = rb[i];
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;
  const bool do_colsum = (p.colsum != nullptr) && tile_n == 0 && tid < BM;

  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    if (ES == 2) {
      const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
      short8_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* pa = As(cur) + (8 * g + q) * PITCH + (wm * 64 + i * 16 + 4 * pp) * 2;
        const char* pb = Bs(cur) + (8 * g + q) * PITCH + (wn * 64 + i * 16 + 4 * pp) * 2;
        short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(pa));
        short4_t a1 =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(pa + 4 * PITCH));
        short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(pb));
        short4_t b1 =
            __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(pb + 4 * PITCH));
        a[i] = short8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        b[i] = short8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
      if (do_colsum) {
        const bf16_t* col = reinterpret_cast<const bf16_t*>(As(cur)) + tid;
#pragma unroll 8
        for (int r = 0; r < BKR; ++r) csum += bf16_to_f32(col[r * (PITCH / 2)]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float a[4], b[4];
        const int rr = 4 * s + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[i] = *reinterpret_cast<const float*>(As(cur) + rr * PITCH + (wm * 64 + i * 16 + (lane & 15)) * 4);
          b[i] = *reinterpret_cast<const float*>(Bs(cur) + rr * PITCH + (wn * 64 + i * 16 + (lane & 15)) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Frag<float>::mma(a[i], b[j], acc[i][j]);
      }
      if (do_colsum) {
        const float* col = reinterpret_cast<const float*>(As(cur)) + tid;
#pragma unroll 8
        for (int r = 0; r < BKR; ++r) csum += col[r * (PITCH / 4)];
      }
    }
    if (kt + 1 < nk) sstore(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long col = n0 + wn * 64 + j * 16 + (lane & 15);
    if (col >= p.No) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row < p.Mo) atomicAdd(p.C + tn_perm(row, p.perm_h) * p.ldc + col, acc[i][j][r]);
      }
  }
  if (do_colsum && m0 + tid < p.Mo) atomicAdd(p.colsum + tn_perm(m0 + tid, p.perm_h), csum);
}

template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_kernel(TnArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * 32 * TROW_BF16];
  gemm_tn_body<T>(p, (int)blockIdx.x, (int)gridDim.x, lds);
}

// grouped launch: one TnArgs descriptor per group (blockIdx.y), e.g. the per-band weight gradients of the band-split
// and mask-decoder 1x1 convolutions (espnet2 BandSplit / MaskDecoder; twin bsrnn_flowse.py:66-81,137-168)
template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_grouped_kernel(const TnArgs* __restrict__ descs) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * 32 * TROW_BF16];
  const TnArgs p = descs[blockIdx.y];
  const int tiles = (int)((p.No + BN - 1) / BN) * (int)((p.Mo + BM - 1) / BM);
  const int nblk = tiles * (int)((p.R + p.rows_per_slice - 1) / p.rows_per_slice);
  if ((int)blockIdx.x >= nblk) return;
  gemm_tn_body<T>(p, (int)blockIdx.x, nblk, lds);
}


}  // namespace urse

using namespace urse;

// ---------------------------------------------------------------------------------------------
// TN, large shapes (bf16): 256 x (32*NTW) output tile per workgroup of 8 waves, 32-row K steps streamed by LDS-DMA
// (global_load_lds_dwordx4, no staging registers) through a 4-stage ring with three stages in flight across the
// barrier (counted vmcnt, raw s_barrier).  Versus the 128x128 kernel above: half the operand bytes per FLOP and
// ~2x the bytes in flight per CU -- that kernel sat at the rate its re-read operands arrive from the Infinity Cache.
// LDS-DMA writes lane-linear (1 KiB per wave-instruction = two 512-byte image rows), so the bank swizzle the
// transposed ds_read_b64_tr_b16 fragment reads need is applied to the per-lane SOURCE column: image segment s (32 B)
// of row r holds global segment s ^ (r & 7); masked rows / columns read a zero page instead (DMA cannot zero-fill).
__device__ uint4 g_tn_zero_page[64];
// perm_h value that selects the TRANSPOSED-output mode of the big TN kernel: C[No, Mo] += (A^T B)^T and `colsum` sums
// the columns of B.  Lets a wide-and-short gradient (fc weight: 196 x 784) be computed as its tall transpose.
constexpr long TN_TRANSPOSED = -(1L << 40);

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to lds_dst + lane*16.  Issued as inline
// asm on purpose: hipcc (ROCm 7.2) drains every outstanding DMA (vmcnt(0)) in front of the next LDS read it can see,
// which serialises the ring; the kernel orders DMA -> read itself with a counted vmcnt and a raw barrier.
// (m0 on a clobber list: clang warns that reserved registers "may not be preserved" - these kernels have no other user of m0)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16(const char* gsrc, char* lds_dst) {
  // the low half of a generic pointer into LDS is the LDS byte address (the aperture sits in the high half): no address-space cast
  // with its null check; m0 is declared clobbered instead of being saved and restored (each cost scalar instructions per DMA, and the
  // scalar unit is shared by the CU's waves)
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// The same transfer through a BUFFER resource: 64 lanes x 16 B from rsrc.base + voff (per lane) to lds_dst + lane*16; a lane whose
// offset is >= rsrc.num_records reads zeros (hardware range check), which replaces every "valid ? pointer : zero page" select.
typedef int rsrc_v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_v4i_t make_rsrc(const char* base, unsigned num_records) {
  const unsigned long a = (unsigned long)base;
  return rsrc_v4i_t{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)num_records, 0x00020000};
}
__device__ __forceinline__ void blds16(unsigned voff, rsrc_v4i_t rsrc, unsigned dst) {      // dst: LDS byte address, wave uniform
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(dst) : "memory");
}
#ifndef URSE_TN_LEAN_ISSUE
#define URSE_TN_LEAN_ISSUE 0   // stage addressing on the scalar unit + buffer range checks instead of per-lane pointer selects:
                               // 103 -> 59 vector and 174 -> 156 scalar instructions per wave and stage, bit-identical gradients,
                               // dual wgrad alone 1.315 -> 1.283 ms (time path) / 1.198 -> 1.086 ms (band path) at 256 workgroups,
                               // but the train step is NOT shorter (same-box A/B, scripts/ab_step_tn_issue.sh: 170.7 / 172.0 ms
                               // against 170.3 / 170.6): like the interleaved-issue variant above, a denser stream in the
                               // kernel that shares CUs with the BPTT costs that kernel what it saves here.  Off.
#endif

// ring depth of the big TN kernel: 5 stages of 32 KB = the whole 160 KB LDS, four stages in flight (measured
// 0.92 vs 1.01 ms for the 3136 x 196 wgrad against 4 stages); the NT kernel is faster with 4 (short K: longer prologue)
#ifndef URSE_TN_PIPE
#define URSE_TN_PIPE 2   // 0 one stage per barrier, 2 two stages per barrier (a half-step software pipeline, reads of the next
                         // half issued under the MFMAs of this one, measured 2.72 vs 2.68 ms and was removed)
                         // (also tried: both stages' fragments read before the first MFMA - 2.53 vs 2.47 ms, dropped)
                         // 4: as 2 with the DMA issue between the MFMA groups - 8 % faster alone (2.23 vs 2.46 ms), but the
                         //    train step is 7 ms SLOWER with it (same-box A/B 189 vs 181 ms): the kernels it shares CUs with
                         //    pay for its denser issue stream; off
#endif
#ifndef URSE_TN_SETPRIO
#define URSE_TN_SETPRIO 1
#endif
#ifndef URSE_TN_NST
#define URSE_TN_NST 5
#endif
#ifndef URSE_NT_SETPRIO
#define URSE_NT_SETPRIO 0   // measured: 1.32 vs 1.23 ms on the gate projection, neutral elsewhere (scripts/abl_nt_prio.py)
#endif
#ifndef URSE_NT_NST
#define URSE_NT_NST 4
#endif
// CSM: 0 no column sums, 1 column sums of A (bias gradient), 2 column sums of B (transposed problem) - a template
// parameter because the unused accumulators would cost 16 / 32 registers of a kernel that sits at the 256 limit
// LDS swizzle key of an image row (32 rows x 512 B, sixteen 32-byte segments per row): one transposed fragment read
// touches the rows 8g + q (g = 0,1 within a 32-lane bank group, q = 0..3) - the key must differ in its low THREE bits
// over those eight rows (a bank is (byte / 4) mod 64, i.e. segment mod 8): q | bit 3 of the row << 2.  (row & 7 put
// rows q and 8 + q on the same banks: every fragment read was a 2-way conflict.)
__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// f16 -> bf16 on a fragment in registers (URSE_BF16_ACT_F16: the weight-gradient GEMMs of an f16-forward training step read the forward's f16
// activations against bf16 gradients; a mixed-operand MFMA does not exist).  v_cvt_f32_f16 (+ sdwa for the high halves) and v_cvt_pk_bf16_f32:
// 12 vector instructions per fragment, issued under the MFMAs - the kernel waits for bytes, not for the vector ALU.  The values differ from the
// bf16 copy the forward kernels used to write beside the f16 one only by double rounding (f32 -> f16 -> bf16 instead of f32 -> bf16).
__device__ __forceinline__ unsigned pair_f16_to_bf16(unsigned v) {      // one dword = two elements: two conversions to f32, one packed conversion back
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(__builtin_convertvector(__builtin_bit_cast(f16x2_, v), f32x2_), bf16x2_));
}
__device__ __forceinline__ short8_t frag_f16_to_bf16(short8_t v) {
  // dword by dword (three instructions and two temporaries each, so that a conversion fits between two MFMAs), the result BUILT from the four dwords:
  // written as `d[k] = f(d[k])` on the vector in an unrolled loop, this compiler (ROCm 7.2 clang 22) converts element 0 twice and replicates it -
  // found by tests/test_gemm_gpu.py, reproduced on a four-line kernel
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ d = __builtin_bit_cast(u32x4_, v);
  const u32x4_ o = u32x4_{pair_f16_to_bf16(d[0]), pair_f16_to_bf16(d[1]), pair_f16_to_bf16(d[2]), pair_f16_to_bf16(d[3])};
  return __builtin_bit_cast(short8_t, o);
}

// CVT: 0 both operands bf16; 1 the A operand is IEEE half (the transposed fc gradient: A = h); 2 the B operand is
template <int NTW, int CSM, int CVT = 0>
__global__ void __launch_bounds__(512) gemm_tn_dma_kernel(TnArgs p) {
  constexpr int BMX = 256, BNX = 32 * NTW, NST = URSE_TN_NST, STAGE = 32768;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const bool dual = p.B2 != nullptr;
  const int tn = dual ? (int)(p.nt1 + (p.No2 + BNX - 1) / BNX) : (int)((p.No + BNX - 1) / BNX);
  const int tiles = tn * (int)((p.Mo + BMX - 1) / BMX);
  const int lid = xcd_remap(blockIdx.x, (int)gridDim.x);
  const int slice = lid / tiles, tile = lid - slice * tiles;
  const int tile_m = tile / tn, tile_n = tile - tile_m * tn;
  const bool second = dual && tile_n >= p.nt1;                 // this tile works on (B2, C2)
  const char* Bop = second ? p.B2 : p.B;
  float* Cop = second ? p.C2 : p.C;
  const long ldb_ = second ? p.ldb2 : p.ldb, ldc_ = second ? p.ldc2 : p.ldc, No_ = second ? p.No2 : p.No;
  const long shift_ = (dual && !second) ? 0 : p.shift, period_ = (dual && !second) ? 0 : p.period;
  const long m0 = (long)tile_m * BMX, n0 = (long)(second ? tile_n - p.nt1 : tile_n) * BNX;
  const long r_begin = (long)slice * p.rows_per_slice;
  long r_end = r_begin + p.rows_per_slice;
  if (r_end > p.R) r_end = p.R;
  if (r_begin >= r_end) return;
  const int nk = (int)((r_end - r_begin + 31) / 32);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;

  // ---- DMA source mapping: wave w fills image rows 4w .. 4w+3 of A and of B (two wave-instructions each) ----
  const char* zsrc = reinterpret_cast<const char*>(g_tn_zero_page) + lane * 16;
  const int half = lane >> 5, chunk = lane & 31;
  long acol[2], bcol[2];
  bool aok[2], bok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rowl = 4 * w + 2 * j + half;
    const int seg = (chunk >> 1) ^ tn_swz(rowl);
    const int cel = seg * 16 + (chunk & 1) * 8;          // element column inside the tile
    acol[j] = m0 + cel;
    bcol[j] = n0 + cel;
    aok[j] = acol[j] < p.Mo;                             // (Mo / No, not the pitch: see gemm_tn_body)
    bok[j] = cel < BNX && bcol[j] < No_;
  }
  // issue() is called for consecutive stages, so each lane's rows, source pointers and the (r / inner) % period phase
  // of the masked operand advance by constants: no multiply / divide in the loop.  (The kernel is issue-bound, not
  // memory-bound: rows pinned into L2 ran no faster.  Moving the row logic to the scalar unit - it is wave-uniform
  // per half-wave - was slower still, 2.92 vs 2.47 ms: one scalar unit serves the CU's eight waves.)
  const int r_end_i = (int)r_end, shift_i = (int)shift_, R_i = (int)p.R;
  const unsigned inner_u = (unsigned)p.inner, per_u = (unsigned)period_, inval_u = (unsigned)p.invalid_step;
  const unsigned step_q = per_u ? (32u / inner_u) % per_u : 0u, step_r = 32u % inner_u;
  const long a_step = 64 * p.lda, b_step = 64 * ldb_;   // bytes per 32 rows
  struct Src {                                           // one of the wave's two (A, B) wave-instructions per stage
    int rr;
    const char* pa;
    const char* pb;
    unsigned ph, rm;
    bool aok, bok;
  };
  Src q0, q1;
  auto init_src = [&](Src& q, int j) __attribute__((always_inline)) {
    q.rr = (int)r_begin + 4 * w + 2 * j + half;
    q.pa = p.A + ((long)q.rr * p.lda + acol[j]) * 2;
    q.pb = Bop + (((long)q.rr + shift_) * ldb_ + bcol[j]) * 2;
    q.ph = per_u ? ((unsigned)q.rr / inner_u) % per_u : 0u;
    q.rm = (unsigned)q.rr % inner_u;
    q.aok = aok[j];
    q.bok = bok[j];
  };
  init_src(q0, 0);
  init_src(q1, 1);
  auto issue_one = [&](Src& q, char* dst) __attribute__((always_inline)) {
    const bool rin = q.rr < r_end_i;                     // (also false for every stage past the slice's last one)
    const int rs = q.rr + shift_i;
    const bool ok = rin & q.bok & ((per_u == 0) | (q.ph != inval_u)) & (rs >= 0) & (rs < R_i);     // (bitwise: no exec-mask branches)
#ifdef TABL_NO_DMA
    asm volatile("" :: "v"(rin && q.aok ? q.pa : zsrc), "v"(ok ? q.pb : zsrc));
#elif defined(TABL_ZERO_DMA)
    glds16(zsrc, dst);
    glds16(zsrc, dst + 16384);
#else
    glds16((rin & q.aok) ? q.pa : zsrc, dst);
    glds16(ok ? q.pb : zsrc, dst + 16384);
#endif
    q.rr += 32;
    q.pa += a_step;
    q.pb += b_step;
    if (per_u) {
      q.rm += step_r;
      const unsigned c = q.rm >= inner_u ? 1u : 0u;
      q.rm -= c ? inner_u : 0u;
      q.ph += step_q + c;
      q.ph -= q.ph >= per_u ? per_u : 0u;
    }
  };
#if URSE_TN_LEAN_ISSUE
  // Lean issue path.  The 32 rows of a stage are addressed from a SCALAR stage base (buffer resource rebuilt per stage on the
  // scalar unit), every lane keeps one constant 32-bit offset per piece, and the hardware range check of the buffer supplies the
  // zeros: rows past the end of the matrix, shifted rows before its start (the offset wraps past num_records), columns outside
  // the operand (offset preset out of range).  Slices are multiples of 32 rows, so a computed stage never crosses its slice
  // end; the stages issued past the last computed one read rows of the next slice into slots nobody reads.  Only the
  // (row / inner) % period mask of the shifted operand is still tracked per lane.  (The generic path below spent ~50 vector and
  // ~100 scalar instructions per wave and stage on per-lane pointer selects; this one spends ~6 + ~30.)
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned lda2 = (unsigned)(p.lda * 2), ldb2 = (unsigned)(ldb_ * 2);
  unsigned voA[2], voB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rowl = 4 * w + 2 * j + half;
    voA[j] = aok[j] ? (unsigned)(((long)rowl * p.lda + acol[j]) * 2) : OOB;
    voB[j] = bok[j] ? (unsigned)(((long)rowl * ldb_ + bcol[j]) * 2) : OOB;
  }
  // stage state, all wave-uniform 32-bit quantities (64-bit compares / multiplies made the scalar path longer than the vector
  // path it replaced): first row of the next stage, its A pointer, and the two resources of the stage being issued
  int st_row = (int)r_begin;
  const char* st_pA = p.A + r_begin * (long)lda2;
  const long stepA = 32L * lda2;
  const int capA = (int)(0x7FFFFFFFu / lda2), capB = (int)(0x7FFFFFFFu / ldb2);
  rsrc_v4i_t cur_rA = make_rsrc(p.A, 0), cur_rB = make_rsrc(Bop, 0);
  unsigned cur_delta = 0;
  auto stage_begin = [&]() __attribute__((always_inline)) {
    int remA = R_i - st_row;
    remA = remA < 0 ? 0 : (remA > capA ? capA : remA);
    cur_rA = make_rsrc(st_pA, (unsigned)remA * lda2);
    const int rowB = st_row + shift_i;
    const int baseRow = rowB < 0 ? 0 : rowB;              // shifted rows before the matrix: base stays at row 0, offsets wrap
    int remB = R_i - baseRow;
    remB = remB < 0 ? 0 : (remB > capB ? capB : remB);
    cur_rB = make_rsrc(Bop + (long)baseRow * ldb2, (unsigned)remB * ldb2);
    cur_delta = (unsigned)((rowB - baseRow) * (int)ldb2);
  };
  const unsigned lds_u = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
  auto issue_lean = [&](Src& q, int j, unsigned dst) __attribute__((always_inline)) {
    if (j == 0) stage_begin();
    blds16(voA[j], cur_rA, dst);
    unsigned vb = voB[j] + cur_delta;
    if (per_u) {
      vb = q.ph != inval_u ? vb : OOB;
      q.rm += step_r;
      const unsigned c = q.rm >= inner_u ? 1u : 0u;
      q.rm -= c ? inner_u : 0u;
      q.ph += step_q + c;
      q.ph -= q.ph >= per_u ? per_u : 0u;
    }
    blds16(vb, cur_rB, dst + 16384);
    if (j == 1) { st_row += 32; st_pA += stepA; }
  };
  auto issue0 = [&](int slot) __attribute__((always_inline)) { issue_lean(q0, 0, lds_u + slot * STAGE + 4 * w * 512); };
  auto issue1 = [&](int slot) __attribute__((always_inline)) { issue_lean(q1, 1, lds_u + slot * STAGE + 4 * w * 512 + 1024); };
#else
  auto issue0 = [&](int slot) __attribute__((always_inline)) { issue_one(q0, lds + slot * STAGE + 4 * w * 512); };
  auto issue1 = [&](int slot) __attribute__((always_inline)) { issue_one(q1, lds + slot * STAGE + 4 * w * 512 + 1024); };
#endif
  auto issue = [&](int slot) __attribute__((always_inline)) {
    issue0(slot);
    issue1(slot);
  };

  f32x4_t acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t accs[CSM == 1 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (CSM == 1 ? 4 : 1); ++i) accs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool trans = p.perm_h == TN_TRANSPOSED;
  const bool do_colsum = CSM == 1 && tile_n == 0 && wn == 0;   // (dual: tile 0 belongs to B)
  const bool do_colsum_b = CSM == 2 && tile_m == 0 && wm == 0;
  f32x4_t accb[CSM == 2 ? NTW : 1];
#pragma unroll
  for (int jj = 0; jj < (CSM == 2 ? NTW : 1); ++jj) accb[jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const short8_t ones = short8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int row0 = 8 * g + q;                           // fragment rows row0 (k 0..3 of the lane) and row0 + 4
  const int off0 = row0 * 512 + pp * 8, off1 = (row0 + 4) * 512 + pp * 8;
  const int sw0 = tn_swz(row0), sw1 = tn_swz(row0 + 4);     // swizzle keys of the two rows

  static_assert(CVT == 0 || (URSE_TN_PIPE == 2 && NST == 5), "mixed operands: default loop form only");
  if constexpr (URSE_TN_PIPE == 4 && NST == 5) {
  // as variant 2, but a stage's DMA issue (address selects, phase update: ~50 VALU instructions per wave) sits BETWEEN
  // the two halves of a stage's MFMAs instead of right behind the barrier, where all eight waves did it at once with the
  // MFMA pipes idle
  // two stages per barrier: the per-k-step barrier cost 200 of 1,000 ns (scripts/abl_tn_parts.py).  At the wait of
  // iteration kt the stages 0 .. kt+2 have been issued and only the youngest may be outstanding (4 DMAs per wave);
  // after the barrier the slots of stages kt-2, kt-1 are free and take stages kt+3, kt+4.
  auto compute = [&](int sl, int nsl) {
    const char* As = lds + sl * STAGE;
    const char* Bs = As + 16384;
    short8_t a[4], b[NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int S = wm * 4 + i;
      short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off0 + ((S ^ sw0) << 5)));
      short4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off1 + ((S ^ sw1) << 5)));
      a[i] = short8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int S = wn * NTW + j;
      short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off0 + ((S ^ sw0) << 5)));
      short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off1 + ((S ^ sw1) << 5)));
      b[j] = short8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    issue0(nsl);
    __builtin_amdgcn_sched_barrier(0);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 2; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __builtin_amdgcn_sched_barrier(0);
    issue1(nsl);
    __builtin_amdgcn_sched_barrier(0);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 4; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (CSM == 2) if (do_colsum_b) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) accb[j] = Frag<bf16_t>::mma(ones, b[j], accb[j]);
    }
    if constexpr (CSM == 1) if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accs[i] = Frag<bf16_t>::mma(a[i], ones, accs[i]);
    }
  };
  issue(0); issue(1); issue(2);
  int slot = 0;
  for (int kt = 0; kt < nk; kt += 2) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int s1 = slot + 1 >= NST ? slot + 1 - NST : slot + 1;
    const int s3 = slot + 3 >= NST ? slot + 3 - NST : slot + 3;
    const int s4 = slot + 4 >= NST ? slot + 4 - NST : slot + 4;
    compute(slot, s3);
    if (kt + 1 < nk) compute(s1, s4);
    else issue(s4);
    slot = slot + 2 >= NST ? slot + 2 - NST : slot + 2;
  }
  } else if constexpr (URSE_TN_PIPE == 2 && NST == 5) {
  // (the only loop form the mixed-operand variants are built for: CVT != 0 is instantiated under the shipping switches)
  // two stages per barrier: the per-k-step barrier cost 200 of 1,000 ns (scripts/abl_tn_parts.py).  At the wait of
  // iteration kt the stages 0 .. kt+2 have been issued and only the youngest may be outstanding (4 DMAs per wave);
  // after the barrier the slots of stages kt-2, kt-1 are free and take stages kt+3, kt+4.
  auto compute = [&](int sl) {
    const char* As = lds + sl * STAGE;
    const char* Bs = As + 16384;
    short8_t a[4], b[NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int S = wm * 4 + i;
      short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off0 + ((S ^ sw0) << 5)));
      short4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off1 + ((S ^ sw1) << 5)));
      a[i] = short8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int S = wn * NTW + j;
      short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off0 + ((S ^ sw0) << 5)));
      short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off1 + ((S ^ sw1) << 5)));
      b[j] = short8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
    if constexpr (CVT == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag_f16_to_bf16(a[i]);
    }
    if constexpr (CVT == 2) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) b[j] = frag_f16_to_bf16(b[j]);
    }
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (CSM == 2) if (do_colsum_b) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) accb[j] = Frag<bf16_t>::mma(ones, b[j], accb[j]);
    }
    if constexpr (CSM == 1) if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accs[i] = Frag<bf16_t>::mma(a[i], ones, accs[i]);
    }
  };
  issue(0); issue(1); issue(2);
  int slot = 0;
  for (int kt = 0; kt < nk; kt += 2) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int s1 = slot + 1 >= NST ? slot + 1 - NST : slot + 1;
    const int s3 = slot + 3 >= NST ? slot + 3 - NST : slot + 3;
    const int s4 = slot + 4 >= NST ? slot + 4 - NST : slot + 4;
    issue(s3);
    issue(s4);
    compute(slot);
    if (kt + 1 < nk) compute(s1);
    slot = slot + 2 >= NST ? slot + 2 - NST : slot + 2;
  }
  } else {
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0) issue(s0);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed for this wave once at most the 4*(NST-2) younger DMAs (stages kt+1 ..) are outstanding; the
    // barrier then makes every wave's part visible and retires the slot that stage kt+NST-1 is about to overwrite
    if (NST == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#ifndef TABL_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    int nslot = slot + NST - 1;
    if (nslot >= NST) nslot -= NST;
    issue(nslot);
    const char* As = lds + slot * STAGE;
    if (++slot == NST) slot = 0;
    const char* Bs = As + 16384;
    short8_t a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int S = wm * 4 + i;
      short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off0 + ((S ^ sw0) << 5)));
      short4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(As + off1 + ((S ^ sw1) << 5)));
      a[i] = short8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#ifdef TABL_NO_READ
      a[i] = ones; asm volatile("" : "+v"(a[i]));
#endif
    }
    short8_t b[NTW];                                  // every fragment read is in flight before the first MFMA
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int S = wn * NTW + j;
      short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off0 + ((S ^ sw0) << 5)));
      short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(Bs + off1 + ((S ^ sw1) << 5)));
      b[j] = short8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#ifdef TABL_NO_READ
      b[j] = ones; asm volatile("" : "+v"(b[j]));
#endif
    }
#ifdef TABL_NO_MFMA
#pragma unroll
    for (int j = 0; j < NTW; ++j) asm volatile("" :: "v"(b[j]));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(a[i]));
#else
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
#endif
    if constexpr (CSM == 2) if (do_colsum_b) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) accb[j] = Frag<bf16_t>::mma(ones, b[j], accb[j]);
    }
    if constexpr (CSM == 1) if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) accs[i] = Frag<bf16_t>::mma(a[i], ones, accs[i]);
    }
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (zero page) DMAs must not outlive the workgroup

#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const long col = n0 + (wn * NTW + j) * 16 + (lane & 15);
    if (col >= No_) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row < p.Mo) {
          if (trans) atomicAdd(Cop + col * ldc_ + row, acc[i][j][r]);
          else atomicAdd(Cop + tn_perm(row, p.perm_h) * ldc_ + col, acc[i][j][r]);
        }
      }
  }
  if constexpr (CSM == 2) if (do_colsum_b && (lane >> 4) == 0) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const long col = n0 + (wn * NTW + j) * 16 + (lane & 15);
      if (col < p.No) atomicAdd(p.colsum + col, accb[j][0]);
    }
  }
  if constexpr (CSM == 1) if (do_colsum && (lane & 15) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row < p.Mo) atomicAdd(p.colsum + tn_perm(row, p.perm_h), accs[i][r]);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// TN, dual operand, 224 x 320 output tile (bf16).  The dual weight gradient of one LSTM direction is [Mo = 4H] x [No | No2] =
// 1568 x (196 | 392): on the 256 x 224 tiles of gemm_tn_dma_kernel that is 7 x 3 tiles of which 23 % is padding (1792 x 672 computed
// for 1568 x 588).  Here the two right-hand operands form ONE virtual matrix [B (No rounded up to 8 columns) | B2 | ones column],
// 200 + 392 + 1 = 593 columns, cut into two 320-column tiles, and the rows into seven 224-row tiles: 14 tiles per row slice, 8 %
// padding.  The ones column turns the bias gradient (column sums of A) into one more output column instead of extra accumulators.
// Workgroup: 8 waves = 2 (m) x 4 (n), 7 x 5 MFMA tiles per wave (140 accumulator registers).  Stage = 32 rows: A image 32 x 512 B
// (224 of 256 columns used), B image columns 0..255 (32 x 512 B) and columns 256..319 (32 x 128 B, its own swizzle key), 36 KB;
// 4-stage LDS-DMA ring, one barrier per stage.  Per lane a DMA source is fixed for the whole slice: B rows, B2 rows (shifted, step
// mask), the ones page or the zero page.
#define URSE_O4 0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u
#define URSE_O16 URSE_O4, URSE_O4, URSE_O4, URSE_O4
#define URSE_O64 URSE_O16, URSE_O16, URSE_O16, URSE_O16
__device__ __attribute__((aligned(1024))) unsigned g_tn_ones_page[256] = {URSE_O64, URSE_O64, URSE_O64, URSE_O64};
#undef URSE_O4
#define URSE_O4 0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u
__device__ __attribute__((aligned(1024))) unsigned g_tn_ones_page_h[256] = {URSE_O64, URSE_O64, URSE_O64, URSE_O64};   // 1.0 in IEEE half (BH)
#undef URSE_O4
#undef URSE_O16
#undef URSE_O64
// swizzle key of the 128-byte-pitch image (four 32-byte segments per row): the eight rows of a 32-lane read group (q, 8 + q) alternate
// between the two bank halves by row parity, so rows of equal parity must differ in the key: bit 1 and bit 3 of the row
__device__ __forceinline__ int tn_swz4(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// One LDS-DMA wave-instruction with the LDS destination given as a byte address in an SGPR (no generic -> LDS pointer cast with its null
// check on the scalar unit, no save / restore of m0: the compiler is told that m0 is clobbered).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void glds16u(const char* gsrc, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// BH: the right-hand operands [B | B2 | ones] are IEEE half (the forward's f16 activations), converted to bf16 behind the fragment read
#ifndef URSE_TN224_CVT_INTERLEAVE
#define URSE_TN224_CVT_INTERLEAVE 1
#endif
#ifdef T224STAMP     // timing diagnostics (scripts/stamps.py): shader-clock stamps of wave 0 of workgroup T224STAMP, [iteration = two stages][8]: 0 top, 1 own DMAs
                     // landed (vmcnt 0), 2 behind the barrier, 3 the next two stages' DMAs issued, 4 first stage computed (fragment reads + 35 MFMAs), 5 second
                     // stage computed.  T224STAMP_SPLIT: 6 = the first stage's fragment reads complete (an lgkmcnt(0) the shipping kernel does not have)
__device__ unsigned long long g_t224stamps[512 * 8];
#define T2ST(slot) do { if (stamp_on && it_ < 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_t224stamps[it_ * 8 + (slot)] = t_; } } while (0)
#else
#define T2ST(slot) do { } while (0)
#endif
template <int DEPTH, bool BH = false>      // stages in flight: 2 (two stages per barrier) or 3 (one barrier per stage).  A template, not a run-time switch: with both loops in one kernel
                          // the register allocation of BOTH got worse (9 spilled registers where round 4's kernel had none) and the second queue's
                          // weight gradients ran 16 % longer - 5.7 ms per train step that hid the round's other gains until a round-over-round A/B
                          // on one box (profiles/r05_ab_round_v2.log)
__global__ void __launch_bounds__(512) gemm_tn_dual224_kernel(TnArgs p) {
  constexpr int BMX = 224, BNX = 320, NST = 4, STAGE = 36864, MT = 7, NT = 5;
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE];
  const long No8 = (p.No + 7) & ~7L;
  const long V = No8 + p.No2;                               // virtual columns [0, No8) from B, [No8, V) from B2, V = ones (colsum)
  const long vones = p.colsum ? V : -1;
  const int tn = (int)((V + (p.colsum ? 1 : 0) + BNX - 1) / BNX);
  const int tiles = tn * (int)(p.Mo / BMX);
  const int lid = xcd_remap(blockIdx.x, (int)gridDim.x);
  const int slice = lid / tiles, tile = lid - slice * tiles;
  const int tile_m = tile / tn, tile_n = tile - tile_m * tn;
  const long m0 = (long)tile_m * BMX, n0 = (long)tile_n * BNX;
  const long r_begin = (long)slice * p.rows_per_slice;
  long r_end = r_begin + p.rows_per_slice;
  if (r_end > p.R) r_end = p.R;
  if (r_begin >= r_end) return;
  const int nk = (int)((r_end - r_begin) / 32);            // the host guarantees whole 32-row stages (R and the slices are multiples of 32)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 2, wn = w & 3;
#ifdef T224STAMP
  const bool stamp_on = blockIdx.x == T224STAMP && w == 0;
  int it_ = 0;
#endif

  // ---- DMA sources.  The issue path is what bounds a ring kernel of this shape (two waves per SIMD, in-order issue): every lane keeps
  // a 64-bit source pointer and a 32-bit step per DMA; a lane without data (padding columns) points at the zero page with step 0, the
  // ones column at the ones page; the only run-time predicate is the step mask of the shifted operand: (row / inner) % period ==
  // invalid_step, tracked per row group.  Rows outside the matrix only occur on masked rows (checked on the host: shift = -/+ inner
  // with the first / last step masked), stages past the slice end are issued from the zero page by a wave-uniform branch.
  const char* zsrc = reinterpret_cast<const char*>(g_tn_zero_page) + lane * 16;
  const char* osrc = reinterpret_cast<const char*>(BH ? g_tn_ones_page_h : g_tn_ones_page) + lane * 16;
  const unsigned inner_u = (unsigned)p.inner, per_u = (unsigned)p.period, inval_u = (unsigned)p.invalid_step;
  const unsigned step_q = (32u / inner_u) % per_u, step_r = 32u % inner_u;
  struct Row { unsigned ph, rm; };                         // (row / inner) % period, row % inner of a DMA row group
  auto init_row = [&](Row& q, int rowl) __attribute__((always_inline)) {
    const unsigned rr = (unsigned)((int)r_begin + rowl);
    q.ph = (rr / inner_u) % per_u;
    q.rm = rr % inner_u;
  };
  auto next_row = [&](Row& q) __attribute__((always_inline)) {
    q.rm += step_r;
    const unsigned c = q.rm >= inner_u ? 1u : 0u;
    q.rm -= c ? inner_u : 0u;
    q.ph += step_q + c;
    q.ph -= q.ph >= per_u ? per_u : 0u;
  };
  struct Src { const char* ptr; unsigned step, bad; };     // bad: the step phase on which the lane reads zeros (~0: never)
  auto init_b = [&](Src& b, long v, int rr0) __attribute__((always_inline)) {
    b.ptr = zsrc; b.step = 0u; b.bad = ~0u;
    if (v < No8) {
      if (v < p.ldb) { b.ptr = p.B + ((long)rr0 * p.ldb + v) * 2; b.step = (unsigned)(64 * p.ldb); }
    } else if (v < V) {
      b.ptr = p.B2 + (((long)rr0 + p.shift) * p.ldb2 + (v - No8)) * 2; b.step = (unsigned)(64 * p.ldb2); b.bad = inval_u;
    } else if (v == (vones & ~7L)) {                       // the 8-column chunk that holds the ones column (the other 7 are dropped)
      b.ptr = osrc;
    }
  };
  const unsigned lds_u = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
  const int half = lane >> 5, chunk = lane & 31;
  Row rw0, rw1, rw2;
  Src sa0, sa1, sb0, sb1, sb2;
  {
    const int rowl0 = 4 * w + half, rowl1 = 4 * w + 2 + half;
    init_row(rw0, rowl0);
    init_row(rw1, rowl1);
    const int cel0 = (((chunk >> 1) ^ tn_swz(rowl0)) << 4) + (chunk & 1) * 8;
    const int cel1 = (((chunk >> 1) ^ tn_swz(rowl1)) << 4) + (chunk & 1) * 8;
    const bool aok0 = cel0 < BMX && m0 + cel0 < p.Mo, aok1 = cel1 < BMX && m0 + cel1 < p.Mo;
    sa0.ptr = aok0 ? p.A + ((long)((int)r_begin + rowl0) * p.lda + m0 + cel0) * 2 : zsrc;
    sa1.ptr = aok1 ? p.A + ((long)((int)r_begin + rowl1) * p.lda + m0 + cel1) * 2 : zsrc;
    sa0.step = aok0 ? (unsigned)(64 * p.lda) : 0u;
    sa1.step = aok1 ? (unsigned)(64 * p.lda) : 0u;
    sa0.bad = sa1.bad = ~0u;
    init_b(sb0, n0 + cel0, (int)r_begin + rowl0);
    init_b(sb1, n0 + cel1, (int)r_begin + rowl1);
    const int rowl2 = 8 * (w & 3) + (lane >> 3), c16 = lane & 7;
    init_row(rw2, rowl2);
    const int cel2 = 256 + (((c16 >> 1) ^ tn_swz4(rowl2)) << 4) + (c16 & 1) * 8;
    init_b(sb2, n0 + cel2, (int)r_begin + rowl2);
  }
  int issued = 0;                                           // stages issued so far (wave-uniform)
  auto issue = [&](int slot) __attribute__((always_inline)) {
    const unsigned st = lds_u + (unsigned)slot * STAGE + (unsigned)w * 2048;
    if (issued < nk) {
#if defined(T224_ZERO_DMA)
      glds16u(zsrc, st); glds16u(zsrc, st + 1024); glds16u(zsrc, st + 16384); glds16u(zsrc, st + 16384 + 1024);
#elif defined(T224_NO_DMA)
      asm volatile("" :: "v"(sa0.ptr), "v"(sa1.ptr), "v"(rw0.ph != sb0.bad ? sb0.ptr : zsrc), "v"(rw1.ph != sb1.bad ? sb1.ptr : zsrc));
#else
      glds16u(sa0.ptr, st);
      glds16u(sa1.ptr, st + 1024);
      glds16u(rw0.ph != sb0.bad ? sb0.ptr : zsrc, st + 16384);
      glds16u(rw1.ph != sb1.bad ? sb1.ptr : zsrc, st + 16384 + 1024);
#endif
      sa0.ptr += sa0.step; sa1.ptr += sa1.step; sb0.ptr += sb0.step; sb1.ptr += sb1.step;
      next_row(rw0);
      next_row(rw1);
      if (w < 4) {
#if defined(T224_ZERO_DMA)
        glds16u(zsrc, lds_u + (unsigned)slot * STAGE + 32768 + (unsigned)w * 1024);
#elif defined(T224_NO_DMA)
        asm volatile("" :: "v"(rw2.ph != sb2.bad ? sb2.ptr : zsrc));
#else
        glds16u(rw2.ph != sb2.bad ? sb2.ptr : zsrc, lds_u + (unsigned)slot * STAGE + 32768 + (unsigned)w * 1024);
#endif
        sb2.ptr += sb2.step;
        next_row(rw2);
      }
    } else {                                                // past the slice: keep the DMA count per stage, read nothing
      glds16u(zsrc, st); glds16u(zsrc, st + 1024); glds16u(zsrc, st + 16384); glds16u(zsrc, st + 16384 + 1024);
      if (w < 4) glds16u(zsrc, lds_u + (unsigned)slot * STAGE + 32768 + (unsigned)w * 1024);
    }
    ++issued;
  };
  // the same stage, one DMA at a time (DEPTH 4: the issue rides between the MFMA groups of the stage being computed).  Round 6 stamps
  // (profiles/r06_stamps_tn224_bwdband_rwx_v1.log): of an iteration's 5,176 cycles the eight waves spend 1,866 ISSUING the next two stages' ten DMAs each
  // right behind the barrier - all at once, blocked on the CU's vector memory path (72 KB at ~39 B/clk), the matrix pipe idle - and 2,276 in MFMAs
  // with that path idle.  part 0 .. 3: the wave's A / B pieces, part 4: the 64-column image's piece (waves 0 .. 3) and the stage's bookkeeping.
  auto issue_part = [&](int slot, int part) __attribute__((always_inline)) {
    const unsigned st = lds_u + (unsigned)slot * STAGE + (unsigned)w * 2048;
    const bool in = issued < nk;
    if (part == 0) glds16u(in ? sa0.ptr : zsrc, st);
    else if (part == 1) glds16u(in ? sa1.ptr : zsrc, st + 1024);
    else if (part == 2) glds16u((in && rw0.ph != sb0.bad) ? sb0.ptr : zsrc, st + 16384);
    else if (part == 3) glds16u((in && rw1.ph != sb1.bad) ? sb1.ptr : zsrc, st + 16384 + 1024);
    else {
      if (w < 4) glds16u((in && rw2.ph != sb2.bad) ? sb2.ptr : zsrc, lds_u + (unsigned)slot * STAGE + 32768 + (unsigned)w * 1024);
      if (in) {
        sa0.ptr += sa0.step; sa1.ptr += sa1.step; sb0.ptr += sb0.step; sb1.ptr += sb1.step;
        next_row(rw0);
        next_row(rw1);
        if (w < 4) { sb2.ptr += sb2.step; next_row(rw2); }
      }
      ++issued;
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment reads (ds_read_b64_tr_b16): rows row0 and row0 + 4 of the image, 8 bytes at pp * 8 of the 32-byte segment S ^ key(row).
  // The segment field (bits 5..8, bits 5..6 in the 128-byte-pitch image) is disjoint from the row / pp bits, so the address is
  // (per-lane constant) ^ (S << 5): one xor-add per read.
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const int row0 = 8 * g + q;
  const unsigned ba0 = (unsigned)(row0 * 512 + pp * 8) ^ ((unsigned)tn_swz(row0) << 5);
  const unsigned ba1 = (unsigned)((row0 + 4) * 512 + pp * 8) ^ ((unsigned)tn_swz(row0 + 4) << 5);
  const unsigned bb0 = (unsigned)(32768 + row0 * 128 + pp * 8) ^ ((unsigned)tn_swz4(row0) << 5);
  const unsigned bb1 = (unsigned)(32768 + (row0 + 4) * 128 + pp * 8) ^ ((unsigned)tn_swz4(row0 + 4) << 5);
  auto frag = [&](unsigned a0, unsigned a1) __attribute__((always_inline)) -> short8_t {
    short4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(size_t)a0);
    short4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(size_t)a1);
    return short8_t{x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  };
  const short8_t ones8 = short8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  (void)ones8;
  const unsigned sA = (unsigned)(wm * MT) << 5, sB = (unsigned)(wn * NT) << 5;
  auto compute_h = [&](int sl) __attribute__((always_inline)) {
    // BH: B fragments are read two ahead and converted one ahead of their MFMAs (f16 -> bf16, three vector instructions per dword between two MFMAs);
    // all five read and converted up front cost the 12 registers this kernel does not have (255 of 256 in the bf16 form)
    const unsigned st = lds_u + (unsigned)sl * STAGE;
    short8_t a[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = frag((ba0 ^ (sA + (i << 5))) + st, (ba1 ^ (sA + (i << 5))) + st);
    // (address selects, not branches: the whole stage stays ONE basic block, which is what lets the scheduler groups below interleave it)
    const bool w3 = wn == 3;
    auto bfrag = [&](int j) __attribute__((always_inline)) -> short8_t {
      const unsigned seg = (w3 ? (j == 0 ? 15u : (unsigned)(j - 1)) : (unsigned)(wn * NT + j)) << 5;
      const bool small = w3 && j > 0;                      // the 64-column image (columns 256 .. 319)
      const unsigned x0 = small ? bb0 : ba0, x1 = small ? bb1 : ba1, off = small ? 0u : 16384u;
      return frag((x0 ^ seg) + st + off, (x1 ^ seg) + st + off);
    };
    short8_t bc = bfrag(0), bn = bfrag(1);
    bc = frag_f16_to_bf16(bc);
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      short8_t bnn = bn;
      if (j + 2 < NT) bnn = bfrag(j + 2);
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], bc, acc[i][j]);
      if (j + 1 < NT) bc = frag_f16_to_bf16(bn);
      bn = bnn;
    }
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if URSE_TN224_CVT_INTERLEAVE
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT + 4, 0);         // the A fragments, b[0], b[1]
    __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);                 // convert b[0]
#pragma unroll
    for (int j = 0; j + 1 < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 6) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // b[j + 1]'s conversion (12) and b[j + 2]'s address selects (~10) spread over the gaps
        if (i == 4 && j + 2 < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // b[j + 2]
      }
    __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
#endif
  };
  auto compute = [&](int sl, int nsl = -1) __attribute__((always_inline)) {
    const unsigned st = lds_u + (unsigned)sl * STAGE;
    short8_t a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#ifdef T224_NO_READ
      a[i] = ones8; asm volatile("" : "+v"(a[i]));
#else
      a[i] = frag((ba0 ^ (sA + (i << 5))) + st, (ba1 ^ (sA + (i << 5))) + st);
#endif
    }
    if (wn < 3) {                                           // segments 0..14: all in the 256-column image
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#ifdef T224_NO_READ
        b[j] = ones8; asm volatile("" : "+v"(b[j]));
#else
        b[j] = frag((ba0 ^ (sB + (j << 5))) + st + 16384, (ba1 ^ (sB + (j << 5))) + st + 16384);
#endif
      }
    } else {                                                // segment 15, then the four segments of the 64-column image
#ifdef T224_NO_READ
#pragma unroll
      for (int j = 0; j < NT; ++j) { b[j] = ones8; asm volatile("" : "+v"(b[j])); }
#else
      b[0] = frag((ba0 ^ (15u << 5)) + st + 16384, (ba1 ^ (15u << 5)) + st + 16384);
#pragma unroll
      for (int j = 1; j < NT; ++j) b[j] = frag((bb0 ^ ((unsigned)(j - 1) << 5)) + st, (bb1 ^ ((unsigned)(j - 1) << 5)) + st);
#endif
    }
#ifdef T224STAMP_SPLIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (sl == 0 || sl == 2) T2ST(6);
#endif
#ifdef T224_NO_MFMA
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(b[j]));
#pragma unroll
    for (int i = 0; i < MT; ++i) asm volatile("" :: "v"(a[i]));
#else
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i][j] = Frag<bf16_t>::mma(a[i], b[j], acc[i][j]);
      if constexpr (DEPTH == 4) {
        if (nsl >= 0) {
          __builtin_amdgcn_sched_barrier(0);
          issue_part(nsl, j);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#if URSE_TN_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#endif
  };
  // two stages per barrier on four slots: at the wait of iteration kt the stages kt, kt + 1 were issued a whole iteration ago; after
  // the barrier the slots of stages kt - 2, kt - 1 are free and take kt + 2, kt + 3, which have two stage times to land
  if constexpr (DEPTH == 3) {
    // THREE stages (108 KB) in flight, one barrier per stage: at the top of iteration kt the stages kt .. kt + 2 are outstanding; the counted
    // wait leaves the two younger ones in flight (this wave's own DMAs: 5 per stage for waves 0-3, 4 for the others), the barrier makes stage kt
    // whole and retires the slot of stage kt - 1, which takes stage kt + 3.  (The two-stages-per-barrier form below drains to ZERO in flight every
    // second stage and has 72 KB in flight at best: at ~1.3 us per stage it is bound by what a CU keeps in flight against the latency of its operands.)
    issue(0);
    issue(1);
    issue(2);
    int slot = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      if (w < 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue((slot + 3) & 3);
      if constexpr (BH) compute_h(slot);
      else compute(slot);
      slot = (slot + 1) & 3;
    }
  } else if constexpr (DEPTH == 4) {
    // two stages per barrier as below, but the DMAs of stages kt + 2 / kt + 3 are issued one by one BETWEEN the MFMA groups of stages kt / kt + 1: a wave
    // blocked on the memory path leaves the matrix pipe to the other wave of its SIMD instead of all eight queueing behind the barrier
    issue(0);
    issue(1);
    int slot = 0;
    for (int kt = 0; kt < nk; kt += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      compute(slot, slot ^ 2);
      if (kt + 1 < nk) compute(slot + 1, (slot ^ 2) + 1);
      slot ^= 2;
    }
  } else {
  issue(0);
  issue(1);
  int slot = 0;
  for (int kt = 0; kt < nk; kt += 2) {
    T2ST(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T2ST(1);
    __builtin_amdgcn_s_barrier();
    T2ST(2);
    issue(slot ^ 2);
    issue((slot ^ 2) + 1);
    T2ST(3);
    const int nh = kt + 1 < nk ? 2 : 1;
#pragma unroll 1
    for (int h = 0; h < nh; ++h) {
      if constexpr (BH) compute_h(slot + h);
      else compute(slot + h);
      T2ST(4 + h);
    }
    slot ^= 2;
#ifdef T224STAMP
    ++it_;
#endif
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (zero page) DMAs must not outlive the workgroup

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const long v = n0 + (wn * NT + j) * 16 + (lane & 15);
    float* dst = nullptr;
    long ld = 0, col = 0;
    if (v < p.No) { dst = p.C; ld = p.ldc; col = v; }
    else if (v >= No8 && v < V) { dst = p.C2; ld = p.ldc2; col = v - No8; }
    else if (v == vones) { dst = p.colsum; ld = 1; col = 0; }
    if (dst == nullptr) continue;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * (MT * 16) + i * 16 + (lane >> 4) * 4 + r;
        atomicAdd(dst + tn_perm(row, p.perm_h) * ld + col, acc[i][j][r]);
      }
  }
}

// ---------------------------------------------------------------------------------------------
static void launch_tn_dma(int ntw, int csm, dim3 grid, hipStream_t st, const TnArgs& p, int cvt = 0) {
  if (cvt == 1) { hipLaunchKernelGGL((gemm_tn_dma_kernel<7, 2, 1>), grid, dim3(512), 0, st, p); return; }     // (the one mixed instance: fc gradient, transposed)
#define URSE_TN_L(N_, C_) hipLaunchKernelGGL((gemm_tn_dma_kernel<N_, C_>), grid, dim3(512), 0, st, p)
  if (ntw == 7) { if (csm == 0) URSE_TN_L(7, 0); else if (csm == 1) URSE_TN_L(7, 1); else URSE_TN_L(7, 2); }
  else { if (csm == 0) URSE_TN_L(8, 0); else if (csm == 1) URSE_TN_L(8, 1); else URSE_TN_L(8, 2); }
#undef URSE_TN_L
}

// NT, large shapes (bf16 operands): 256 x (32*NTW) tile per workgroup of 8 waves (4 x 2, 64 x 16*NTW each), K in
// 32-element steps on the same 4-stage LDS-DMA ring as gemm_tn_dma_kernel.  Image rows are 64 bytes (4 chunks of
// 16 B); physical chunk = k-chunk ^ ((row >> 1) & 3), applied on the DMA source address and on the ds_read_b128
// fragment reads, so the 16 rows of a fragment hit 16 different 16-byte bank groups.  Epilogue as gemm_nt_kernel
// (bias / tanh / tanh-backward / residual, output staged through LDS for 16-byte coalesced stores).
struct NtExtra {       // optional outputs of the ring NT kernel beside C (null / 0 = none)
  double* gstats;      // GroupNorm statistics of C per group of `rpg` rows: [groups][2] (sum, sum of squares), pre-zeroed
  long rpg;
  // GroupNorm BACKWARD sums of C (= dy, the gradient w.r.t. the normalised tensor) against the normalised input x (GNB kernels only):
  //   gnb_sums[g] += (sum dy * gamma, sum dy * gamma * xhat) over group g (rpg rows x N), xhat = (x - mean_g) * rstd_g from gnb_stats;
  //   gnb_part[slot][0][c] += sum dy * xhat, gnb_part[slot][1][c] += sum dy per channel c (slot = workgroup % gnb_slots: the 1,704
  //   workgroups of a C2 launch would serialise on 2 N addresses otherwise, norm.hip "red_elems"); a fold kernel adds the slots up
  const float* gnb_x;
  const double* gnb_stats;
  const float* gnb_gamma;
  double* gnb_sums;
  float* gnb_part;
  int gnb_slots;
  float gnb_eps;
};

template <typename TO, int NTW, int ACT, int BMX, int WNC, int GNB = 0, typename TI = bf16_t>
__device__ __forceinline__ void gemm_nt_dma_body(const GemmDesc& d, double* gstats = nullptr, long rpg = 0, const NtExtra* xt = nullptr) {
  // tile = BMX rows x (16 * NTW * WNC) columns; waves BMX/64 (m) x WNC (n), each 64 x 16*NTW
  // <256, 2>: 8 waves, 256 x 224/256, 4 stages of 32 KB (long K: least operand traffic per FLOP)
  // <128, 4>: 8 waves, 128 x 448, 4 stages of 36 KB (short K, write-bound outputs: 896-byte row segments reach 5.4 TB/s
  //           of HBM write rate where 448-byte segments reach 3.7, scripts/diag/write_pattern.py)
  // <128, 2>: 4 waves, 128 x 224, 3 stages of 24 KB, two workgroups per CU
  constexpr int BNX = 16 * NTW * WNC, NWV = BMX / 64 * WNC, NTHR = NWV * 64;
  constexpr int BNR = WNC == 2 ? 256 : 16 * NTW * WNC;                     // B image rows
  constexpr int NST = NWV == 8 ? URSE_NT_NST : 3, STAGE = (BMX + BNR) * 64, BOFF = BMX * 64;
  constexpr int AI = (BMX / 16 + NWV - 1) / NWV;  // A wave-instructions per wave per stage
  constexpr int BI = (BNR / 16 + NWV - 1) / NWV;  // B wave-instructions per wave per stage (the surplus ones read the zero page)
  constexpr int DPW = AI + BI;                    // DMAs per wave per stage
  static_assert(AI * NWV * 16 == BMX, "A image blocks divide evenly over the waves");
  __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE + 1024];   // + one block for surplus (zero page) DMAs
  const int tn = (int)((d.N + BNX - 1) / BNX), tm = (int)((d.M + BMX - 1) / BMX);
  if ((int)blockIdx.x >= tm * tn) return;            // grouped launch: grid.x is the largest group's tile count
  const int bid = xcd_remap(blockIdx.x, tm * tn);
  const int tile_m = bid / tn, tile_n = bid - tile_m * tn;
  const long m0 = (long)tile_m * BMX, n0 = (long)tile_n * BNX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w / WNC, wn = w % WNC;
  const int nk = (int)(d.K / 32);
  const char* zsrc = reinterpret_cast<const char*>(g_tn_zero_page) + lane * 16;
  const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 3) & 3);     // source k-chunk of this lane's slot
  const int lc = lane & 15, lr = lane >> 4;
  // fragment byte offset inside a 16-row block.  Swizzle key (row >> 1) & 3: ds_read_b128 is served in the four 16-lane
  // groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 (MI355X_MICROARCH.md, LDS), and only this key spreads each group
  // over all 64 banks ((row >> 2) & 3, used before, left every read a 2-way conflict - same as no swizzle at all)
  const int foff = lc * 64 + ((lr ^ ((lc >> 1) & 3)) << 4);

  // DMA: wave w fills the 16-row blocks AI*w .. of the A image and BI*w .. of the B image
  const char* pa[AI];
  const char* pb[BI];
  bool aok[AI], bok[BI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int rowl = 16 * (AI * w + j) + srow;
    aok[j] = rowl < BMX && m0 + rowl < d.M;
    pa[j] = d.A + ((m0 + rowl) * d.lda) * 2 + schunk * 16;
  }
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int rowl = 16 * (BI * w + j) + srow;
    bok[j] = rowl < BNX && n0 + rowl < d.N;
    pb[j] = d.B + ((n0 + rowl) * d.ldb) * 2 + schunk * 16;
  }
  auto issue = [&](int kt, int slot) {
    char* sbase = lds + slot * STAGE;
    const bool kin = kt < nk;
#pragma unroll
    for (int j = 0; j < AI; ++j) glds16((kin && aok[j]) ? pa[j] + (long)kt * 64 : zsrc, sbase + (AI * w + j) * 1024);
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int blk = BI * w + j;          // a surplus block (keeps the per-wave DMA count uniform) lands in the spare block
      glds16((kin && bok[j]) ? pb[j] + (long)kt * 64 : zsrc, blk < BNR / 16 ? sbase + BOFF + blk * 1024 : lds + NST * STAGE);
    }
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s, s);

  constexpr int OS = sizeof(TO);
  // The weight fragment is passed as the MFMA "A" operand, so the accumulator tile is D[n][m]: lane (lr, lc) holds
  // output row m = lc and the FOUR CONSECUTIVE columns n = 4*lr .. 4*lr+3 -> 8-byte (bf16) / 16-byte (f32) pieces of an
  // output row per lane, staged with one LDS write per tile instead of four 2-byte ones.
  f32x4_t bv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const long col = n0 + (wn * NTW + j) * 16 + lr * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = (d.bias && col + r < d.N) ? d.bias[col + r] : 0.f;
  }
  f32x4_t acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed for this wave once only the (NST-2) younger stages are outstanding; the barrier makes every
    // wave's part visible and retires the slot that stage kt+NST-1 is about to overwrite
    if ((NST - 2) * DPW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if ((NST - 2) * DPW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if ((NST - 2) * DPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    static_assert((NST - 2) * DPW == 12 || (NST - 2) * DPW == 10 || (NST - 2) * DPW == 8 || (NST - 2) * DPW == 6, "wait count");
    __builtin_amdgcn_s_barrier();
    int nslot = slot + NST - 1;
    if (nslot >= NST) nslot -= NST;
    issue(kt + NST - 1, nslot);
    const char* As = lds + slot * STAGE + wm * 4096 + foff;
    const char* Bs = lds + slot * STAGE + BOFF + wn * NTW * 1024 + foff;
    short8_t a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const short8_t*>(As + i * 1024);
    short8_t b[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) b[j] = *reinterpret_cast<const short8_t*>(Bs + j * 1024);
#if URSE_NT_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = Frag<TI>::mma(b[j], a[i], acc[i][j]);   // D[n][m]: see the epilogue
#if URSE_NT_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (++slot == NST) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef NABL_NO_EPI
  if (acc[0][0][0] == 123.456f) d.C[0] = 1;
  return;
#endif

  // ---- epilogue: the tile goes through LDS (the ring is idle) RPP rows at a time, all waves of those rows writing
  // at once, so that global stores / residual reads are 16-byte coalesced ----
  constexpr int CP = BNX * OS + 16;
  constexpr int RPP = (BMX * CP <= NST * STAGE) ? BMX : ((BMX / 2 * CP <= NST * STAGE) ? BMX / 2 : BMX / 4);
  static_assert(RPP * CP <= NST * STAGE && RPP >= 64, "staging does not fit");
  constexpr int EPC = 16 / OS;
  constexpr int CPR = BNX / EPC;
  TO* C = reinterpret_cast<TO*>(d.C);
  // f16 forward mode, ACT 1: the aux slot names a second output, the same values in bf16 (see gemm_nt_kernel)
  constexpr bool C2K = __is_same(TI, f16_t) && OS == 2 && ACT == 1;
  bf16_t* C2 = (C2K && d.resid) ? reinterpret_cast<bf16_t*>(const_cast<float*>(d.resid)) : nullptr;
  const bool vec_ok = ((d.ldc * OS) % 16 == 0) && ((reinterpret_cast<uintptr_t>(d.C) % 16) == 0) &&
                      (!d.resid || ACT == 2 || (((d.ldr * 4) % 16 == 0) && (reinterpret_cast<uintptr_t>(d.resid) % 16) == 0)) &&
                      (!C2 || (d.ldr * 2) % 16 == 0);
  // (storing straight from the accumulators, 8/16 B per lane, measured 2.3x slower than this staged epilogue)
  // optional GroupNorm statistics of the OUTPUT (f32, after the residual): sum and sum of squares per group of `rpg` consecutive
  // rows, accumulated from the values the sweep below stores - the separate pass over the tensor (gn_stats_kernel) goes away.
  // A tile spans at most two groups (rpg >= BMX, checked by the launcher).
  float gsa = 0.f, gqa = 0.f, gsb = 0.f, gqb = 0.f;
  const bool gst = OS == 4 && gstats != nullptr;
  const long gidx = gst ? m0 / rpg : 0, gbound = gst ? (gidx + 1) * rpg : 0;
  // GNB: group means / rstds of the (at most two) groups of this tile, the thread's four gammas, per-channel partial sums
  float gnb_m[2] = {0.f, 0.f}, gnb_r[2] = {0.f, 0.f}, gnb_ga[4] = {0.f, 0.f, 0.f, 0.f}, gnb_dg[4] = {0.f, 0.f, 0.f, 0.f}, gnb_db[4] = {0.f, 0.f, 0.f, 0.f};
  long gnb_g0 = 0, gnb_bound = 0;
  // x rows of a pass: the first GXA requested in front of the staging (their HBM latency hides behind it), the rest behind the barrier, in front
  // of the sweep that consumes the first ones - all XI at once sat on top of the second pass's 112 live accumulators: 6 registers to scratch
  constexpr int GXI = GNB ? (RPP + NTHR / CPR - 1) / (NTHR / CPR) : 1, GXA = GXI < 8 ? GXI : 8;
  float4 gnb_xpre[GXI];
  if constexpr (GNB) {
    gnb_g0 = m0 / xt->rpg;
    gnb_bound = (gnb_g0 + 1) * xt->rpg;
    const double cnt = (double)xt->rpg * (double)d.N;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const long g = gnb_g0 + q;
      if (g * xt->rpg < d.M) {
        const double mu = xt->gnb_stats[g * 2] / cnt;
        double var = xt->gnb_stats[g * 2 + 1] / cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        gnb_m[q] = (float)mu;
        gnb_r[q] = (float)(1.0 / sqrt(var + (double)xt->gnb_eps));
      }
    }
    const int chf = tid % (BNX / (16 / OS));
    const long col = n0 + chf * (16 / OS);
#pragma unroll
    for (int q = 0; q < 4; ++q) gnb_ga[q] = (col + q < d.N) ? xt->gnb_gamma[col + q] : 0.f;
  }
#pragma unroll 1
  for (int pass = 0; pass < BMX / RPP; ++pass) {
    if (pass > 0) __syncthreads();
    if constexpr (GNB) {
      constexpr int SLOTS = NTHR / CPR, XI = (RPP + SLOTS - 1) / SLOTS;
      const int slot = tid / CPR, chf = tid - slot * CPR;
      const long col = n0 + chf * EPC;
      static_assert(XI == GXI, "prefetch array");
#pragma unroll
      for (int i = 0; i < GXA; ++i) {
        const int lr_ = slot + i * SLOTS;
        const long row = m0 + pass * RPP + lr_;
        const bool ok = slot < SLOTS && col < d.N && lr_ < RPP && row < d.M;
        gnb_xpre[i] = ok ? *reinterpret_cast<const float4*>(xt->gnb_x + row * d.ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if ((wm * 64) / RPP == pass) {
      const int rbase = (wm * 64) % RPP;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int lcol = (wn * NTW + j) * 16 + lr * 4;
        const long col = n0 + lcol;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int lrow = rbase + i * 16 + lc;
          f32x4_t v = acc[i][j] + bv[j];
          if (ACT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
          }
          if (ACT == 2) {   // tanh backward: aux (= resid slot, TO typed) holds h = tanh(.)
            const long row = m0 + wm * 64 + i * 16 + lc;
            const TO* hp = reinterpret_cast<const TO*>(d.resid) + row * d.ldr + col;
            float hv[4] = {0.f, 0.f, 0.f, 0.f};
            // the lane's four columns are consecutive: one 8-byte load instead of four 2-byte ones (the scalar form made this
            // epilogue 6x the kernel's byte floor on the mask decoder's backward)
            if constexpr (OS == 2) {
              if (row < d.M && col + 3 < d.N && ((reinterpret_cast<uintptr_t>(hp) & 7) == 0)) {
                const uint2 q = *reinterpret_cast<const uint2*>(hp);
                unpack2<TO>(q.x, hv[0], hv[1]);
                unpack2<TO>(q.y, hv[2], hv[3]);
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (row < d.M && col + r < d.N) hv[r] = to_f32<TO>(hp[r]);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (row < d.M && col + r < d.N) hv[r] = to_f32<TO>(hp[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= (1.f - hv[r] * hv[r]);
          }
          char* dst = lds + lrow * CP + lcol * OS;
          if constexpr (OS == 2) {
            uint2 pk;
            pk.x = pack2<TO>(v[0], v[1]);
            pk.y = pack2<TO>(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = pk;
          } else {
            *reinterpret_cast<f32x4_t*>(dst) = v;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (GNB) {
      // fixed column chunk per thread (NTHR / CPR row slots), so the per-channel sums stay in registers; a tile spans at most two groups.
      // The thread's x values of this pass were requested before the staging (gnb_xpre): their HBM latency is behind it
      constexpr int SLOTS = NTHR / CPR, XI = (RPP + SLOTS - 1) / SLOTS;
      const int slot = tid / CPR, chf = tid - slot * CPR;
      const long col = n0 + chf * EPC;
#pragma unroll
      for (int i = GXA; i < XI; ++i) {
        const int lr_ = slot + i * SLOTS;
        const long row = m0 + pass * RPP + lr_;
        const bool ok = slot < SLOTS && col < d.N && lr_ < RPP && row < d.M;
        gnb_xpre[i] = ok ? *reinterpret_cast<const float4*>(xt->gnb_x + row * d.ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (slot < SLOTS && col < d.N) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const int lr_ = slot + i * SLOTS;
          const long row = m0 + pass * RPP + lr_;
          if (lr_ < RPP && row < d.M) {
            const float4 f = *reinterpret_cast<const float4*>(lds + lr_ * CP + chf * 16);
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(&f), reinterpret_cast<f32x4_t*>(C + row * d.ldc + col));
            const float4 xv = gnb_xpre[i];
            const bool first = row < gnb_bound;
            const float mean = first ? gnb_m[0] : gnb_m[1], rstd = first ? gnb_r[0] : gnb_r[1];
            const float dy[4] = {f.x, f.y, f.z, f.w}, xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float xh = (xs[q] - mean) * rstd;
              gnb_dg[q] += dy[q] * xh;
              gnb_db[q] += dy[q];
              a1 += dy[q] * gnb_ga[q];
              a2 += dy[q] * gnb_ga[q] * xh;
            }
            if (first) { gsa += a1; gqa += a2; } else { gsb += a1; gqb += a2; }
          }
        }
      }
      continue;
    }
    // the threads sweep RPP rows x CPR 16-byte chunks; (lrow, ch) advance incrementally (no division in the loop)
    int lrow = tid / CPR, ch = tid - lrow * CPR;
    constexpr int DROW = NTHR / CPR, DCH = NTHR - DROW * CPR;
    for (; lrow < RPP; lrow += DROW) {
      const long row = m0 + pass * RPP + lrow, col = n0 + ch * EPC;
      if (row < d.M && col < d.N) {
        const char* src = lds + lrow * CP + ch * 16;
        if (vec_ok && col + EPC <= d.N) {
          uint4 v = *reinterpret_cast<const uint4*>(src);
          if (OS == 4 && d.resid && ACT != 2) {
            const float4 rr = *reinterpret_cast<const float4*>(d.resid + row * d.ldr + col);
            float4 f = *reinterpret_cast<float4*>(&v);
            f.x += rr.x; f.y += rr.y; f.z += rr.z; f.w += rr.w;
            v = *reinterpret_cast<uint4*>(&f);
          }
          if (gst) {
            const float4 f = *reinterpret_cast<const float4*>(&v);
            const float sv = (f.x + f.y) + (f.z + f.w), qv = (f.x * f.x + f.y * f.y) + (f.z * f.z + f.w * f.w);
            if (row < gbound) { gsa += sv; gqa += qv; } else { gsb += sv; gqb += qv; }
          }
#ifdef NABL_NO_GST
          if (v.x == 0x12345678u)
#endif
          // streaming output (read next by a different kernel): non-temporal store, measured -16 % on the [M, 8H] gates
          __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(&v), reinterpret_cast<f32x4_t*>(C + row * d.ldc + col));
          if constexpr (C2K) {
            if (C2) {
              float a0, a1, a2, a3, a4, a5, a6, a7;
              unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
              const uint4 o = make_uint4(pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7));
              __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(&o), reinterpret_cast<f32x4_t*>(C2 + row * d.ldr + col));
            }
          }
        } else {
          const TO* sv = reinterpret_cast<const TO*>(src);
          for (int e = 0; e < EPC && col + e < d.N; ++e) {
            float f = to_f32<TO>(sv[e]);
            if (OS == 4 && d.resid && ACT != 2) f += d.resid[row * d.ldr + col + e];
            if (gst) {
              if (row < gbound) { gsa += f; gqa += f * f; } else { gsb += f; gqb += f * f; }
            }
            C[row * d.ldc + col + e] = from_f32<TO>(f);
            if constexpr (C2K) {
              if (C2) C2[row * d.ldr + col + e] = f32_to_bf16(f);
            }
          }
        }
      }
      ch += DCH;
      if (ch >= CPR) { ch -= CPR; ++lrow; }
    }
  }
  if constexpr (GNB) {
    __syncthreads();                                           // the staging area is free again
    constexpr int SLOTS = NTHR / CPR;
    double* red = reinterpret_cast<double*>(lds);
    float4* pg = reinterpret_cast<float4*>(lds + 1024);       // [SLOTS][CPR] sum dy * xhat, then [SLOTS][CPR] sum dy
    float4* pb = pg + SLOTS * CPR;
    const double r0 = wave_sum_d((double)gsa), r1 = wave_sum_d((double)gqa), r2 = wave_sum_d((double)gsb), r3 = wave_sum_d((double)gqb);
    if (lane == 0) { red[w * 4] = r0; red[w * 4 + 1] = r1; red[w * 4 + 2] = r2; red[w * 4 + 3] = r3; }
    if (tid < SLOTS * CPR) {
      pg[tid] = make_float4(gnb_dg[0], gnb_dg[1], gnb_dg[2], gnb_dg[3]);
      pb[tid] = make_float4(gnb_db[0], gnb_db[1], gnb_db[2], gnb_db[3]);
    }
    __syncthreads();
    if (tid < 4) {
      double t = 0.0;
      for (int i = 0; i < NWV; ++i) t += red[i * 4 + tid];
      const long g = gnb_g0 + (tid >> 1);
      if ((tid < 2 || gnb_bound < m0 + BMX) && g * xt->rpg < d.M) atomicAdd(xt->gnb_sums + g * 2 + (tid & 1), t);
    }
    if (tid >= 64 && tid < 64 + CPR) {
      const int chf = tid - 64;
      const long col = n0 + chf * EPC;
      if (col < d.N) {
        float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
        for (int i = 0; i < SLOTS; ++i) {
          const float4 a = pg[i * CPR + chf], bq = pb[i * CPR + chf];
          sg.x += a.x; sg.y += a.y; sg.z += a.z; sg.w += a.w;
          sb.x += bq.x; sb.y += bq.y; sb.z += bq.z; sb.w += bq.w;
        }
        float* part = xt->gnb_part + (long)(blockIdx.x % xt->gnb_slots) * 2 * d.N;
        const float vg[4] = {sg.x, sg.y, sg.z, sg.w}, vb[4] = {sb.x, sb.y, sb.z, sb.w};
        for (int q = 0; q < 4 && col + q < d.N; ++q) {
          atomicAdd(part + col + q, vg[q]);
          atomicAdd(part + d.N + col + q, vb[q]);
        }
      }
    }
    return;
  }
  if (gst) {
    __syncthreads();                                           // the staging area is free again
    double* red = reinterpret_cast<double*>(lds);
    const double r0 = wave_sum_d((double)gsa), r1 = wave_sum_d((double)gqa), r2 = wave_sum_d((double)gsb), r3 = wave_sum_d((double)gqb);
    if (lane == 0) { red[w * 4] = r0; red[w * 4 + 1] = r1; red[w * 4 + 2] = r2; red[w * 4 + 3] = r3; }
    __syncthreads();
    if (tid < 4) {
      double t = 0.0;
      for (int i = 0; i < NWV; ++i) t += red[i * 4 + tid];
      const long g = gidx + (tid >> 1);
      if (tid < 2 || gbound < m0 + BMX) {                      // the second pair only when the tile reaches into the next group
        if (g * rpg < d.M) atomicAdd(gstats + g * 2 + (tid & 1), t);
      }
    }
  }
}

template <typename TO, int NTW, int ACT, int BMX, int WNC = 2, typename TI = bf16_t>
__global__ void __launch_bounds__(BMX / 64 * WNC * 64) gemm_nt_dma_kernel(GemmDesc d, NtExtra x) {
  gemm_nt_dma_body<TO, NTW, ACT, BMX, WNC, 0, TI>(d, x.gstats, x.rpg);
}

// f32 output + the GroupNorm-backward sums of that output (see NtExtra): the dgrad GEMM whose result feeds urse_groupnorm_bwd_apply
__global__ void __launch_bounds__(512) gemm_nt_dma_gnb_kernel(GemmDesc d, NtExtra x) {
  gemm_nt_dma_body<float, 7, 0, 256, 2, 1>(d, nullptr, 0, &x);
}

// dgamma[c] += sum over slots part[s][0][c], dbeta[c] += sum over slots part[s][1][c]
__global__ void gnb_fold_kernel(const float* __restrict__ part, int slots, int N, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < slots; ++s) { a += part[(long)s * 2 * N + c]; b += part[(long)s * 2 * N + N + c]; }
  dgamma[c] += a;
  dbeta[c] += b;
}

// grouped form (one descriptor per band, blockIdx.y = group): the per-band 1x1 convolutions of the mask decoder /
// band split at B*T = 12,832 rows per band
template <typename TO, int ACT, typename TI = bf16_t>
__global__ void __launch_bounds__(512) gemm_nt_dma_grouped_kernel(const GemmDesc* __restrict__ descs) {
  const GemmDesc d = descs[blockIdx.y];
  gemm_nt_dma_body<TO, 7, ACT, 256, 2, 0, TI>(d);
}

// B-stationary NT for short K and wide N (the gate projection: K = 224, N = 3136, bf16 out).  In the ring kernel above a
// 128 x 448 tile re-loads 200 KB of (L2-resident) weights for 57 KB of activations, and a workgroup's LDS-DMA ring
// sustains only ~30 GB/s: the weight re-loads, not the output, set its 1.36 ms.  Here a workgroup keeps its 224 weight
// rows x K resident in LDS (98 KB) and walks 256-row tiles of M; only the activations stream, through a 3-stage ring of
// 16 KB, which is also the epilogue's staging area (64 rows per pass).  Workgroups of different column slices take the
// same M tiles in the same order, so an activation tile comes from HBM once and from L2 / the Infinity Cache after.
template <int ACT, typename TI = bf16_t>      // TI: operand AND output format (bf16 | f16)
__global__ void __launch_bounds__(512) gemm_nt_bres_kernel(GemmDesc d, int wg_per_slice) {
  constexpr int BMX = 256, NTW = 7, BNX = 224, NST = 3, STAGE = BMX * 64, KSTEPS = 7, BRES = KSTEPS * 14 * 1024;
  __shared__ __attribute__((aligned(1024))) char lds[BRES + NST * STAGE];
  char* ring = lds + BRES;
  const int tm = (int)((d.M + BMX - 1) / BMX);
  const int tile_n = blockIdx.x / wg_per_slice, part = blockIdx.x - tile_n * wg_per_slice;
  const long n0 = (long)tile_n * BNX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
  const int nk = (int)(d.K / 32);
  const char* zsrc = reinterpret_cast<const char*>(g_tn_zero_page) + lane * 16;
  const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 3) & 3);
  const int lc = lane & 15, lr = lane >> 4;
  const int foff = lc * 64 + ((lr ^ ((lc >> 1) & 3)) << 4);
  // resident weights: block (kt, b) = the 16 rows 16b .. of k-step kt, same image as a ring stage's B part
  for (int blk = w; blk < nk * 14; blk += 8) {
    const int kt = blk / 14, b = blk - kt * 14;
    const long row = n0 + b * 16 + srow;
    glds16(row < d.N ? d.B + (row * d.ldb) * 2 + schunk * 16 + (long)kt * 64 : zsrc, lds + blk * 1024);
  }
  f32x4_t bv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const long col = n0 + (wn * NTW + j) * 16 + lr * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = (d.bias && col + r < d.N) ? d.bias[col + r] : 0.f;
  }
  bf16_t* C = reinterpret_cast<bf16_t*>(d.C);
  // stage j of a tile lives in ring slot (j + 2) % 3: slot 2 is the one the epilogue's staging (slots 0-1) leaves alone, so the
  // NEXT tile's first stage is fetched into it while this tile is being written out
  auto tile_src = [&](int mt, const char* (&pa)[2], bool (&aok)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long row = (long)mt * BMX + 16 * (2 * w + j) + srow;
      aok[j] = mt < tm && row < d.M;
      pa[j] = d.A + (row * d.lda) * 2 + schunk * 16;
    }
  };
  auto issue = [&](const char* (&pa)[2], bool (&aok)[2], int kt, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      glds16((kt < nk && aok[j]) ? pa[j] + (long)kt * 64 : zsrc, ring + slot * STAGE + (2 * w + j) * 1024);
  };
  const char* pa[2];
  bool aok[2];
  tile_src(part, pa, aok);
  issue(pa, aok, 0, 2);
  for (int mt = part; mt < tm; mt += wg_per_slice) {
    const long m0 = (long)mt * BMX;
    issue(pa, aok, 1, 0);
    f32x4_t acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int slot = 2;                                        // slot of stage kt
    for (int kt = 0; kt < nk; ++kt) {
      // all but this wave's two newest DMAs (stage kt+1) have landed: stage kt, the previous tile's stores and, the first
      // time, the resident weights
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int nslot = slot + 2;                              // stage kt+2 -> the slot stage kt-1 has left
      if (nslot >= NST) nslot -= NST;
      issue(pa, aok, kt + 2, nslot);
      const char* As = ring + slot * STAGE + wm * 4096 + foff;
      const char* Bs = lds + kt * (14 * 1024) + wn * NTW * 1024 + foff;
      short8_t a[4], b[NTW];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const short8_t*>(As + i * 1024);
#pragma unroll
      for (int j = 0; j < NTW; ++j) b[j] = *reinterpret_cast<const short8_t*>(Bs + j * 1024);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = Frag<TI>::mma(b[j], a[i], acc[i][j]);   // D[n][m], as in the ring kernel
      if (++slot == NST) slot = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the trailing zero-page stages included)
    __syncthreads();
    tile_src(mt + wg_per_slice, pa, aok);
    issue(pa, aok, 0, 2);                                // next tile's first stage, in flight under the epilogue
    // epilogue: 64 rows per pass through slots 0-1 of the (idle) ring, 16-byte non-temporal stores; LDS-only barriers, so
    // neither the prefetch nor the stores are waited for here
    constexpr int CP = BNX * 2 + 16, RPP = 64, CPR = BNX / 8, DROW = 512 / CPR, DCH = 512 - DROW * CPR;
    static_assert(RPP * CP <= 2 * STAGE, "staging must leave slot 2 alone");
#pragma unroll 1
    for (int pass = 0; pass < BMX / RPP; ++pass) {
      if (pass > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (wm == pass) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          const int lcol = (wn * NTW + j) * 16 + lr * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4_t v = acc[i][j] + bv[j];
            if (ACT == 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
            }
            uint2 pk;
            pk.x = pack2<TI>(v[0], v[1]);
            pk.y = pack2<TI>(v[2], v[3]);
            *reinterpret_cast<uint2*>(ring + (i * 16 + lc) * CP + lcol * 2) = pk;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int lrow = tid / CPR, ch = tid - lrow * CPR;
      for (; lrow < RPP; lrow += DROW) {
        const long row = m0 + pass * RPP + lrow, col = n0 + ch * 8;
        if (row < d.M && col < d.N) {
          const char* src = ring + lrow * CP + ch * 16;
          if (col + 8 <= d.N) {
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4_t*>(src), reinterpret_cast<f32x4_t*>(C + row * d.ldc + col));
          } else {
            const bf16_t* sv = reinterpret_cast<const bf16_t*>(src);
            for (int e = 0; e < 8 && col + e < d.N; ++e) C[row * d.ldc + col + e] = sv[e];
          }
        }
        ch += DCH;
        if (ch >= CPR) { ch -= CPR; ++lrow; }
      }
    }
    // slots 0-1 are staging no more once every wave has read its share: the next tile's second stage may land
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the last (zero-page) prefetch must not outlive the workgroup
}

#ifdef URSE_EXPERIMENTS      // round 4's register-stationary gate projection: faster alone, no faster in the step - variant builds only
// Weights resident in REGISTERS, for K = 224 and wide N with bf16 output (the time path's gate projection: N = 3,136).
// gemm_nt_bres_kernel keeps a 224-column weight slice in LDS, so a workgroup writes 448-byte pieces of an output row, and 448-byte
// segments cap the HBM write rate at 3.7 TB/s where 896-byte segments reach 5.4 (scripts/diag/write_pattern.py) - the kernel is bound by its
// 2.74 GB of output.  Here a workgroup owns 448 COLUMNS: seven compute waves hold 64 columns each as MFMA operand fragments (4 column
// tiles x 7 k-slabs x 4 VGPRs = 112 registers, loaded once), an eighth wave streams 32-row activation stages (14 KB, the ring GEMMs'
// swizzled 64-byte-row image) through a 5-slot LDS ring with four stages in flight, one raw barrier per stage (the row-wave LSTM's
// protocol: at barrier k stage k has landed and the slot of stage k - 1 is free).  A stage's 32 x 448 outputs go to one of two LDS
// staging tiles and leave during the NEXT stage as 16-byte pieces of 896-byte row segments.  Same k order per output element as the
// ring / LDS-resident kernels: bit-identical results.
constexpr int WR_ROWS = 32, WR_KS = 7, WR_NSLOT = 7, WR_STAGE = WR_KS * 2 * 1024, WR_BNX = 448, WR_CP = WR_BNX * 2 + 16, WR_AHEAD = WR_NSLOT - 1;
template <int ACT>
__global__ void __launch_bounds__(448) gemm_nt_wreg_kernel(GemmDesc d, int parts, long rows_per_part) {
  __shared__ __attribute__((aligned(1024))) char lds[WR_NSLOT * WR_STAGE + 2 * WR_ROWS * WR_CP];
  char* outs = lds + WR_NSLOT * WR_STAGE;
  // the column slices of one row part take consecutive ids on ONE XCD (xcd_remap): its activation rows come from HBM once and from that XCD's L2
  // for the other slices
  const int tn_ = (int)gridDim.x / parts;
  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int part = lid / tn_, tile_n = lid - part * tn_;
  const long n0 = (long)tile_n * WR_BNX;
  const long r_begin = (long)part * rows_per_part;
  long r_end = r_begin + rows_per_part;
  if (r_end > d.M) r_end = d.M;
  if (r_begin >= r_end) return;
  const int nst = (int)((r_end - r_begin + WR_ROWS - 1) / WR_ROWS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, lr = lane >> 4;
  // activation stages: stage s = rows r_begin + 32 s ..; image block (ks, half) = 16 rows x 64 B of k-slab ks, chunk swizzled as in the ring
  // GEMMs; wave w brings k-slab w of every stage (two DMAs); a row at or beyond r_end lies outside the buffer resource and arrives as zeros
  const int srow = lane >> 2, schunk = (lane & 3) ^ ((lane >> 3) & 3);
  const rsrc_v4i_t ra = make_rsrc(d.A, (unsigned)(r_end * d.lda * 2));
  const unsigned lds0 = (unsigned)(size_t)lds;
  auto issue = [&](int s) __attribute__((always_inline)) {
    const int slot = s % WR_NSLOT;
    const long row0 = r_begin + (long)s * WR_ROWS + srow;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const long row = row0 + hf * 16;
#ifdef WRABL_NO_DMA
      const unsigned voff = (s < 0 && row < r_end) ? (unsigned)((row * d.lda + w * 32) * 2 + schunk * 16) : 0xFFFFF000u;
#else
      const unsigned voff = (s < nst && row < r_end) ? (unsigned)((row * d.lda + w * 32) * 2 + schunk * 16) : 0xFFFFF000u;
#endif
      blds16(voff, ra, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(slot * WR_STAGE + (w * 2 + hf) * 1024)));
    }
  };
  const long nw = n0 + w * 64;                                          // this wave's 64 columns
  short8_t wreg[4][WR_KS];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long n = nw + j * 16 + lc;
#pragma unroll
    for (int ks = 0; ks < WR_KS; ++ks)
      wreg[j][ks] = n < d.N ? *reinterpret_cast<const short8_t*>(d.B + (n * d.ldb + ks * 32 + 8 * lr) * 2) : short8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  f32x4_t bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long col = nw + j * 16 + lr * 4 + r;
      bv[j][r] = (d.bias && col < d.N) ? d.bias[col] : 0.f;
    }
  bf16_t* C = reinterpret_cast<bf16_t*>(d.C);
  const int foff = lc * 64 + ((lr ^ ((lc >> 1) & 3)) << 4);
  constexpr int CPR = WR_BNX / 8, NTHC = 7 * 64, NPC = WR_ROWS * CPR / NTHC;     // 56 pieces of 16 B per output row, 4 per thread and stage
  static_assert(WR_ROWS * CPR % NTHC == 0, "the threads divide a stage's output pieces evenly");
  auto sweep = [&](int k) __attribute__((always_inline)) {              // the staged outputs of stage k -> global, 16-byte pieces along the rows
    const char* ob = outs + (k & 1) * (WR_ROWS * WR_CP);
    const long row0 = r_begin + (long)k * WR_ROWS;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(d.C, 0, (int)0xFFFFF000u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int idx = tid + i * NTHC, lrow = idx / CPR, ch = idx - lrow * CPR;
      const long row = row0 + lrow, col = n0 + ch * 8;
      const uint4 v = *reinterpret_cast<const uint4*>(ob + lrow * WR_CP + ch * 16);
      // one buffer store per piece whatever its fate (the counted vmcnt below relies on NPC stores per stage): a piece that must not be
      // stored gets an offset outside the buffer; the ragged last 16 bytes of a row (N % 8 != 0) are written element by element instead
#ifdef WRABL_NO_STORE      // timing diagnostics (wrong results): WRABL_NO_STORE, WRABL_NO_DMA, WRABL_NO_MFMA
      const bool whole = false && row < r_end && col + 8 <= d.N;
#else
      const bool whole = row < r_end && col + 8 <= d.N;
#endif
      const unsigned off = whole ? (unsigned)((row * d.ldc + col) * 2) : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rc, (int)off, 0, 2);      // (aux 2 = nt / slc: streaming output)
      if (!whole && row < r_end && col < d.N) {
        const bf16_t* sv = reinterpret_cast<const bf16_t*>(ob + lrow * WR_CP + ch * 16);
        for (int e = 0; e < 8 && col + e < d.N; ++e) C[row * d.ldc + col + e] = sv[e];
      }
    }
  };
#pragma unroll
  for (int s = 0; s < WR_AHEAD; ++s) issue(s);
  // stage 0: everything but the five younger stages' DMAs of this wave (and, before them in the queue, the weight fragments) has landed
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int k = 0; k < nst; ++k) {
    issue(k + WR_AHEAD);                                                // into the slot of stage k - 1, free behind the barrier just passed
    if (k > 0) sweep(k - 1);                                            // complete behind that barrier; the stores overlap this stage's MFMAs
    const char* st = lds + (k % WR_NSLOT) * WR_STAGE + foff;
    f32x4_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < WR_KS; ++ks) {
      short8_t x[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) x[i] = *reinterpret_cast<const short8_t*>(st + (ks * 2 + i) * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
#ifdef WRABL_NO_MFMA
        for (int i = 0; i < 2; ++i) acc[i][j][ks & 3] += __builtin_bit_cast(float, (int)x[i][0] | ((int)wreg[j][ks][1] << 16));
#else
        for (int i = 0; i < 2; ++i) acc[i][j] = Frag<bf16_t>::mma(wreg[j][ks], x[i], acc[i][j]);     // D[n][m]: lane (lr, lc) = row lc, columns 4 lr ..
#endif
    }
    char* ob = outs + (k & 1) * (WR_ROWS * WR_CP);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4_t v = acc[i][j] + bv[j];
        if (ACT == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf_(v[r]);
        }
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(ob + (i * 16 + lc) * WR_CP + (w * 64 + j * 16 + lr * 4) * 2) = pk;
      }
    // this wave's part of stage k + 1 has landed: behind its two DMAs the queue holds the DMAs of stages k + 2 .. k + 6 (10) and the NPC
    // stores of each sweep since (iterations 1 .. k, at most the last six count)
    {
      const int kk = k < 6 ? k : 6;
      switch (kk) {
        case 0: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
      }
    }
    static_assert(WR_AHEAD == 6 && NPC == 4, "the counted waits above are written for six stages ahead and four stores per sweep");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // stage k + 1 has landed; every wave's outputs of stage k are staged
  }
  sweep(nst - 1);
}
#endif   // URSE_EXPERIMENTS

static int check_desc_host(const GemmDesc& d, int es, const char* who) {
  URSE_CHECK_ARG(d.A && d.B && d.C, "%s: null operand", who);
  URSE_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "%s: empty problem", who);
  URSE_CHECK_ARG((d.K * es) % SLAB == 0, "%s: K (%ld) must be a multiple of %d elements", who, d.K, SLAB / es);
  URSE_CHECK_ARG((d.lda * es) % 16 == 0 && (d.ldb * es) % 16 == 0 && ((uintptr_t)d.A % 16) == 0 &&
                     ((uintptr_t)d.B % 16) == 0,
                 "%s: operands must be 16-byte aligned with 16-byte-multiple row pitch", who);
  URSE_CHECK_ARG(d.lda >= d.K && d.ldb >= d.K && d.ldc >= d.N, "%s: leading dimension too small", who);
  return URSE_OK;
}

template <typename T, typename TO>
static void launch_nt(const GemmDesc* descs, const GemmDesc& single, int groups, int max_blocks, int act,
                      hipStream_t st) {
  hipLaunchKernelGGL((gemm_nt_kernel<T, TO>), dim3(max_blocks, 1, groups), dim3(256), 0, st, descs, single, act);
}

static int dispatch_nt(const GemmDesc* descs, const GemmDesc& single, int groups, int max_blocks, int in_dtype,
                       int out_dtype, int act, hipStream_t st) {
  if (in_dtype == URSE_BF16 && out_dtype == URSE_BF16) launch_nt<bf16_t, bf16_t>(descs, single, groups, max_blocks, act, st);
  else if (in_dtype == URSE_BF16 && out_dtype == URSE_F32) launch_nt<bf16_t, float>(descs, single, groups, max_blocks, act, st);
  else if (in_dtype == URSE_F32 && out_dtype == URSE_F32) launch_nt<float, float>(descs, single, groups, max_blocks, act, st);
  else if (in_dtype == URSE_F32 && out_dtype == URSE_BF16) launch_nt<float, bf16_t>(descs, single, groups, max_blocks, act, st);
  else if (in_dtype == URSE_F16 && out_dtype == URSE_F16) launch_nt<f16_t, f16_t>(descs, single, groups, max_blocks, act, st);
  else if (in_dtype == URSE_F16 && out_dtype == URSE_F32) launch_nt<f16_t, float>(descs, single, groups, max_blocks, act, st);
  else { set_error("gemm_nt: bad dtype %d/%d", in_dtype, out_dtype); return URSE_ERR_INVALID_ARG; }
  URSE_CHECK_LAUNCH("urse_gemm_nt");
  return URSE_OK;
}

// measured (scripts/abl_nt_wide.py): 128x448 tiles win only on the [M, 8H] gate projection (K=224, bf16 out: 1.32 -> 1.22 ms);
// at N=800 / K>=512 / f32 out the 256x224 tile stays ahead
static long g_nt_bres_min_n = getenv("URSE_NT_BRES_MIN_N") ? atol(getenv("URSE_NT_BRES_MIN_N")) : 448;   // (448 also takes the fc dgrad, N = 784: 0.37 -> 0.28 ms alone; in the step 135.43 against 135.83 ms, mean of six interleaved same-box runs each, profiles/r04_ab_nt_bres_fc_dgrad_v1.log; round 3 had measured it 0.9 ms slower beside that round's second queue)
static int g_nt_bres_wgs = 256;      // persistent workgroups of the weight-stationary NT kernel (one per CU)
static int g_nt_wide_default = 1;
static long g_nt_wide_maxk = 256;

static int gemm_nt_impl(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                        const float* bias, const float* resid, int64_t ldr, int64_t M, int64_t N, int64_t K,
                        int in_dtype, int out_dtype, int act, void* stream, double* gstats, long rpg, int* fused) {
  NtExtra xtra{};
  GemmDesc d;
  d.A = (const char*)A; d.B = (const char*)B; d.C = (char*)C; d.bias = bias; d.resid = resid;
  d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.M = M; d.N = N; d.K = K; d.ldr = ldr;
  const bool in16 = in_dtype == URSE_BF16 || in_dtype == URSE_F16;      // 16-bit operands: the same kernels, the MFMA of the format
  const bool f16 = in_dtype == URSE_F16;
  URSE_CHECK_ARG(in16 || in_dtype == URSE_F32, "urse_gemm_nt: bad operand dtype %d", in_dtype);
  URSE_CHECK_ARG(out_dtype == URSE_F32 || (out_dtype == URSE_BF16 && !f16) || (out_dtype == URSE_F16 && f16),
                 "urse_gemm_nt: output dtype %d with operand dtype %d (16-bit outputs take the operands' format)", out_dtype, in_dtype);
  const bool out16 = out_dtype != URSE_F32;
  int rc = check_desc_host(d, in16 ? 2 : 4, "urse_gemm_nt");
  if (rc) return rc;
  URSE_CHECK_ARG(!resid || act == 2 || out_dtype == URSE_F32 || (in_dtype == URSE_F16 && out_dtype == URSE_F16 && act == 1),
                 "urse_gemm_nt: residual epilogue writes f32 (f16 operands with act 1: the aux slot is the bf16 copy of the output)");
  URSE_CHECK_ARG(act != 2 || resid, "urse_gemm_nt: act 2 (tanh backward) needs the aux operand");
  static const bool no_dma = getenv("URSE_NT_NO_DMA") != nullptr;
  const char* bres_env = getenv("URSE_NT_BRES");
  const int bres_mode = bres_env ? atoi(bres_env) : 1;
#ifdef URSE_EXPERIMENTS
  // OFF by default (0): 1.11 -> 0.80 ms per launch alone (scripts/bench_gemm.py), but 1.01 ms in the step's profile against 0.96 for the
  // LDS-resident kernel, and no difference in the step (136.1 / 135.1 vs 136.3 / 135.1 ms, profiles/r04_exp_nt_wreg_v1.log): with one barrier
  // per 32-row stage its MFMA phase (0.33 ms, seven waves on four SIMDs) is in series with the memory phases.  Read per call: tests switch it.
  const long wreg_min_n = getenv("URSE_NT_WREG_MIN_N") ? atol(getenv("URSE_NT_WREG_MIN_N")) : 0;
  if (wreg_min_n > 0 && in_dtype == URSE_BF16 && out_dtype == URSE_BF16 && !no_dma && M >= 8192 && N >= wreg_min_n && K == 32 * WR_KS &&
      !resid && act != 2 && (ldc * 2) % 16 == 0 && ((uintptr_t)C % 16) == 0 && M * lda * 2 < 0xFFFFF000L && M * ldc * 2 < 0xFFFFF000L) {
    // weights resident in registers, 448-column slices: see gemm_nt_wreg_kernel
    const int tn = (int)((N + WR_BNX - 1) / WR_BNX);
    int parts = 252 / tn;
    if (parts < 1) parts = 1;
    long rpp = (M + parts - 1) / parts;
    rpp = (rpp + WR_ROWS - 1) / WR_ROWS * WR_ROWS;
    dim3 grid((unsigned)(tn * parts));
    hipStream_t st = (hipStream_t)stream;
    note_launch(URSE_KV_NT_BRES);
    if (act == 0) hipLaunchKernelGGL(gemm_nt_wreg_kernel<0>, grid, dim3(448), 0, st, d, parts, rpp);
    else hipLaunchKernelGGL(gemm_nt_wreg_kernel<1>, grid, dim3(448), 0, st, d, parts, rpp);
    URSE_CHECK_LAUNCH("urse_gemm_nt");
    return URSE_OK;
  }
#endif
  if (bres_mode && in16 && out16 && !no_dma && M >= 8192 && N >= g_nt_bres_min_n && K % 32 == 0 &&
      K >= 96 && K <= 224 && !resid && act != 2 && (ldc * 2) % 16 == 0 && ((uintptr_t)C % 16) == 0) {
    // weight-stationary tiles: see gemm_nt_bres_kernel
    const int tn = (int)((N + 223) / 224);
    int per = g_nt_bres_wgs / tn;
    if (per < 1) per = 1;
    dim3 grid((unsigned)(tn * per));
    hipStream_t st = (hipStream_t)stream;
    note_launch(URSE_KV_NT_BRES);
    if (f16) {
      if (act == 0) hipLaunchKernelGGL((gemm_nt_bres_kernel<0, f16_t>), grid, dim3(512), 0, st, d, per);
      else hipLaunchKernelGGL((gemm_nt_bres_kernel<1, f16_t>), grid, dim3(512), 0, st, d, per);
    } else if (act == 0) hipLaunchKernelGGL(gemm_nt_bres_kernel<0>, grid, dim3(512), 0, st, d, per);
    else hipLaunchKernelGGL(gemm_nt_bres_kernel<1>, grid, dim3(512), 0, st, d, per);
    URSE_CHECK_LAUNCH("urse_gemm_nt");
    return URSE_OK;
  }
  if (in16 && !no_dma && M >= 2048 && N >= 160 && K % 32 == 0 && K >= 96 && !(f16 && act == 2)) {
    const long pad7 = (N + 223) / 224 * 224, pad8 = (N + 255) / 256 * 256;
    int ntw = pad7 <= pad8 ? 7 : 8;
    if (const char* e = getenv("URSE_NT_NTW")) ntw = atoi(e) == 8 ? 8 : 7;
    int bmx = 256;                        // (the 128-row, two-workgroups-per-CU variant measured no faster at any K)
    if (const char* e = getenv("URSE_NT_BMX")) bmx = atoi(e) == 128 ? 128 : 256;
    // 128 x 448 tiles (896-byte output row segments) for the write-bound shapes: short K, wide N
    int wide = g_nt_wide_default && K <= g_nt_wide_maxk && N >= 1792 && out_dtype == URSE_BF16 && (N + 447) / 448 * 448 <= pad7 + 64;
    if (const char* e = getenv("URSE_NT_WIDE")) wide = atoi(e) != 0 && N >= 448;
    if (f16) { wide = 0; bmx = 256; }        // (the f16 forward mode instantiates the 256-row tile only)
    const long tl = wide ? ((M + 127) / 128) * ((N + 447) / 448) : ((M + bmx - 1) / bmx) * ((N + 32L * ntw - 1) / (32L * ntw));
    URSE_CHECK_ARG(tl < (1L << 31), "urse_gemm_nt: too many tiles");
    dim3 grid((unsigned)tl);
    hipStream_t st = (hipStream_t)stream;
    if (gstats && out_dtype == URSE_F32 && act != 2 && !wide && rpg >= bmx) {   // statistics of C ride on the epilogue sweep
      xtra.gstats = gstats;
      xtra.rpg = rpg;
      if (fused) *fused = 1;
    }
#define URSE_NT_DMA(TO_, NTW_, ACT_, BMX_) \
  hipLaunchKernelGGL((gemm_nt_dma_kernel<TO_, NTW_, ACT_, BMX_>), grid, dim3(BMX_ * 2), 0, st, d, xtra)
#define URSE_NT_DMA_W(TO_, ACT_) \
  hipLaunchKernelGGL((gemm_nt_dma_kernel<TO_, 7, ACT_, 128, 4>), grid, dim3(512), 0, st, d, xtra)
#define URSE_NT_DMA_B(TO_, NTW_, ACT_) do { if (wide) URSE_NT_DMA_W(TO_, ACT_); else if (bmx == 128) URSE_NT_DMA(TO_, NTW_, ACT_, 128); else URSE_NT_DMA(TO_, NTW_, ACT_, 256); } while (0)
#define URSE_NT_DMA_ACT(TO_, NTW_) \
  do { if (act == 0) URSE_NT_DMA_B(TO_, NTW_, 0); else if (act == 1) URSE_NT_DMA_B(TO_, NTW_, 1); else URSE_NT_DMA_B(TO_, NTW_, 2); } while (0)
    note_launch(wide ? URSE_KV_NT_RING_WIDE : URSE_KV_NT_RING);
    if (f16) {
#define URSE_NT_DMA_H(TO_, NTW_, ACT_) hipLaunchKernelGGL((gemm_nt_dma_kernel<TO_, NTW_, ACT_, 256, 2, f16_t>), grid, dim3(512), 0, st, d, xtra)
#define URSE_NT_DMA_H_ACT(TO_, NTW_) do { if (act == 0) URSE_NT_DMA_H(TO_, NTW_, 0); else URSE_NT_DMA_H(TO_, NTW_, 1); } while (0)
      if (out_dtype == URSE_F16) { if (ntw == 7) URSE_NT_DMA_H_ACT(f16_t, 7); else URSE_NT_DMA_H_ACT(f16_t, 8); }
      else { if (ntw == 7) URSE_NT_DMA_H_ACT(float, 7); else URSE_NT_DMA_H_ACT(float, 8); }
    } else if (out_dtype == URSE_BF16) {
      if (ntw == 7) URSE_NT_DMA_ACT(bf16_t, 7); else URSE_NT_DMA_ACT(bf16_t, 8);
    } else {
      if (ntw == 7) URSE_NT_DMA_ACT(float, 7); else URSE_NT_DMA_ACT(float, 8);
    }
    URSE_CHECK_LAUNCH("urse_gemm_nt");
    return URSE_OK;
  }
  const long blocks = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  URSE_CHECK_ARG(blocks < (1L << 31), "urse_gemm_nt: too many tiles");
  note_launch(URSE_KV_NT_128);
  return dispatch_nt(nullptr, d, 1, (int)blocks, in_dtype, out_dtype, act, (hipStream_t)stream);
}

extern "C" int urse_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                            const float* bias, const float* resid, int64_t ldr, int64_t M, int64_t N, int64_t K,
                            int in_dtype, int out_dtype, int act, void* stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, bias, resid, ldr, M, N, K, in_dtype, out_dtype, act, stream, nullptr, 0, nullptr);
}

// urse_gemm_nt with f32 output + the GroupNorm statistics of that output: stats[g] = (sum, sum of squares) over rows
// [g * rows_per_group, (g + 1) * rows_per_group) x all N columns, the layout urse_groupnorm_apply / _bwd read.  Fused into the
// ring kernel's epilogue when that kernel takes the shape; otherwise the statistics pass runs behind the GEMM.
extern "C" int urse_gemm_nt_gnstats(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                    const float* bias, const float* resid, int64_t ldr, int64_t M, int64_t N, int64_t K,
                                    int in_dtype, int act, double* stats, int64_t rows_per_group, void* stream) {
  URSE_CHECK_ARG(stats && rows_per_group > 0 && M % rows_per_group == 0 && ldc == N && N % 4 == 0,
                 "urse_gemm_nt_gnstats: needs whole groups of rows, a dense output and N %% 4 == 0");
  const int groups = (int)(M / rows_per_group);
  (void)hipMemsetAsync(stats, 0, sizeof(double) * 2 * groups, (hipStream_t)stream);
  int fused = 0;
  int rc = gemm_nt_impl(A, lda, B, ldb, C, ldc, bias, resid, ldr, M, N, K, in_dtype, URSE_F32, act, stream, stats, rows_per_group,
                        &fused);
  if (rc || fused) return rc;
  return urse_groupnorm_stats(C, stats, groups, (int)rows_per_group, 1, (int)N, (int)N, stream);
}

// C[M, N] (f32, dense) = A[M, K] @ B[N, K]^T, and - from the tile while it is on the chip - the sums the GroupNorm backward needs of C = dy
// against the tensor x that was normalised (same [M, N] layout; groups of rows_per_group rows; `stats` as urse_groupnorm_fwd wrote them):
//   sums[g] = (sum dy * gamma, sum dy * gamma * xhat) (f64, overwritten), dgamma[c] += sum dy * xhat, dbeta[c] += sum dy.
// Replaces the reduce pass of urse_groupnorm_bwd (one read of x and of dy: 684 MB per half layer at C2); urse_groupnorm_bwd_apply takes the sums.
// part: workspace of slots * 2 * N floats.  URSE_ERR_UNSUPPORTED when the shape does not run on the ring kernel - the caller then
// uses urse_gemm_nt + urse_groupnorm_bwd.   (espnet2 BSRNN: the LayerNorm/GroupNorm in front of every LSTM, baseline twin bsrnn_flowse.py:296-306)
extern "C" int urse_gemm_nt_gnbwd(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t M, int64_t N, int64_t K,
                                  int in_dtype, const float* x, const double* stats, const float* gamma, double* sums, float* dgamma,
                                  float* dbeta, float* part, int slots, int64_t rows_per_group, float eps, void* stream) {
  URSE_CHECK_ARG(A && B && C && x && stats && gamma && sums && dgamma && dbeta && part && slots > 0 && rows_per_group > 0,
                 "urse_gemm_nt_gnbwd: bad argument");
  static const bool no_dma = getenv("URSE_NT_NO_DMA") != nullptr;
  if (no_dma || in_dtype != URSE_BF16 || M < 2048 || N < 160 || N > 224 || N % 4 || K % 32 || K < 96 || rows_per_group < 256 ||
      M % rows_per_group || ((uintptr_t)C % 16) || ((uintptr_t)x % 16)) {
    set_error("urse_gemm_nt_gnbwd: shape M%ld N%ld K%ld not served by the fused kernel", (long)M, (long)N, (long)K);
    return URSE_ERR_UNSUPPORTED;
  }
  GemmDesc d;
  d.A = (const char*)A; d.B = (const char*)B; d.C = (char*)C; d.bias = nullptr; d.resid = nullptr;
  d.lda = lda; d.ldb = ldb; d.ldc = N; d.M = M; d.N = N; d.K = K; d.ldr = 0;
  int rc = check_desc_host(d, 2, "urse_gemm_nt_gnbwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t sums_bytes = sizeof(double) * 2 * (size_t)(M / rows_per_group), part_bytes = sizeof(float) * 2 * (size_t)N * slots;
  if (reinterpret_cast<char*>(part) == reinterpret_cast<char*>(sums) + sums_bytes) {      // one workspace, sums then part: one fill
    (void)hipMemsetAsync(sums, 0, sums_bytes + part_bytes, st);
  } else {
    (void)hipMemsetAsync(sums, 0, sums_bytes, st);
    (void)hipMemsetAsync(part, 0, part_bytes, st);
  }
  NtExtra xt{};
  xt.rpg = rows_per_group; xt.gnb_x = x; xt.gnb_stats = stats; xt.gnb_gamma = gamma; xt.gnb_sums = sums; xt.gnb_part = part;
  xt.gnb_slots = slots; xt.gnb_eps = eps;
  const long tl = (M + 255) / 256;
  URSE_CHECK_ARG(tl < (1L << 31), "urse_gemm_nt_gnbwd: too many tiles");
  note_launch(URSE_KV_NT_RING);
  hipLaunchKernelGGL(gemm_nt_dma_gnb_kernel, dim3((unsigned)tl), dim3(512), 0, st, d, xt);
  hipLaunchKernelGGL(gnb_fold_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, part, slots, (int)N, dgamma, dbeta);
  URSE_CHECK_LAUNCH("urse_gemm_nt_gnbwd");
  return URSE_OK;
}

extern "C" int urse_gemm_nt_grouped(const void* descs, int groups, int max_blocks, int in_dtype, int out_dtype,
                                    int act, void* stream) {
  URSE_CHECK_ARG(descs && groups > 0 && max_blocks > 0, "urse_gemm_nt_grouped: bad argument");
  GemmDesc dummy;
  memset(&dummy, 0, sizeof(dummy));
  note_launch(URSE_KV_NT_GROUPED_128);
  return dispatch_nt((const GemmDesc*)descs, dummy, groups, max_blocks, in_dtype, out_dtype, act,
                     (hipStream_t)stream);
}

// grouped form with a host mirror of the descriptors: the library validates them, sizes the grid and picks the kernel
// (LDS-DMA ring when every group is bf16 with K % 32 == 0 and N >= 160, the 128 x 128 kernel otherwise)
extern "C" int urse_gemm_nt_grouped_h(const void* descs, const int64_t* host_descs, int groups, int in_dtype,
                                      int out_dtype, int act, void* stream) {
  URSE_CHECK_ARG(descs && host_descs && groups > 0 && groups < 65536, "urse_gemm_nt_grouped_h: bad argument");
  const GemmDesc* hd = reinterpret_cast<const GemmDesc*>(host_descs);
  const bool f16 = in_dtype == URSE_F16;
  const int es = (in_dtype == URSE_BF16 || f16) ? 2 : 4;
  URSE_CHECK_ARG(out_dtype == URSE_F32 || (out_dtype == URSE_BF16 && !f16) || (out_dtype == URSE_F16 && f16),
                 "urse_gemm_nt_grouped_h: output dtype %d with operand dtype %d", out_dtype, in_dtype);
  bool dma = es == 2 && !(f16 && act == 2) && !getenv("URSE_NT_NO_DMA") && !getenv("URSE_NT_GROUPED_NO_DMA");
  long t128 = 0, tdma = 0;
  const char* mm = getenv("URSE_NT_GROUPED_MIN_M");      // (tests lower it to run the ring kernel on small batches)
  const long min_m = mm ? atol(mm) : 1024;
  for (int g = 0; g < groups; ++g) {
    const GemmDesc& d = hd[g];
    if (int rc = check_desc_host(d, es, "urse_gemm_nt_grouped_h")) return rc;
    dma = dma && d.K % 32 == 0 && d.N >= 160 && d.M >= min_m;
    t128 = std::max(t128, ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN));
    tdma = std::max(tdma, ((d.M + 255) / 256) * ((d.N + 223) / 224));
  }
  URSE_CHECK_ARG(t128 < (1L << 31), "urse_gemm_nt_grouped_h: too many tiles");
  if (!dma) {
    GemmDesc dummy;
    memset(&dummy, 0, sizeof(dummy));
    note_launch(URSE_KV_NT_GROUPED_128);
    return dispatch_nt((const GemmDesc*)descs, dummy, groups, (int)t128, in_dtype, out_dtype, act, (hipStream_t)stream);
  }
  dim3 grid((unsigned)tdma, (unsigned)groups);
  hipStream_t st = (hipStream_t)stream;
  const GemmDesc* dd = (const GemmDesc*)descs;
#define URSE_NT_G(TO_, ACT_) hipLaunchKernelGGL((gemm_nt_dma_grouped_kernel<TO_, ACT_>), grid, dim3(512), 0, st, dd)
#define URSE_NT_G_ACT(TO_) do { if (act == 0) URSE_NT_G(TO_, 0); else if (act == 1) URSE_NT_G(TO_, 1); else URSE_NT_G(TO_, 2); } while (0)
  note_launch(URSE_KV_NT_GROUPED_RING);
#define URSE_NT_GH(TO_, ACT_) hipLaunchKernelGGL((gemm_nt_dma_grouped_kernel<TO_, ACT_, f16_t>), grid, dim3(512), 0, st, dd)
  if (f16 && out_dtype == URSE_F16) { if (act == 0) URSE_NT_GH(f16_t, 0); else URSE_NT_GH(f16_t, 1); }
  else if (f16) { if (act == 0) URSE_NT_GH(float, 0); else URSE_NT_GH(float, 1); }
  else if (out_dtype == URSE_BF16) URSE_NT_G_ACT(bf16_t);
  else if (out_dtype == URSE_F32) URSE_NT_G_ACT(float);
  else { set_error("urse_gemm_nt_grouped_h: bad output dtype %d", out_dtype); return URSE_ERR_INVALID_ARG; }
  URSE_CHECK_LAUNCH("urse_gemm_nt_grouped_h");
  return URSE_OK;
}

// target_workgroups: workgroups the big TN kernels aim for (0 = 256, one per CU).  A caller that runs them beside another
// kernel on part of the chip (bsrnn.py: deferred wgrads next to the time path's BPTT) passes the CUs that are actually free,
// so the launch is one round of long workgroups instead of two-and-a-bit rounds.  Per call: the library keeps no tuning state.

extern "C" int urse_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                            float* colsum, int64_t R, int64_t Mo, int64_t No, int64_t shift, int64_t inner,
                            int64_t period, int64_t invalid_step, int64_t perm_h, int dtype, int target_workgroups,
                            void* stream) {
  URSE_CHECK_ARG(A && B && C && R > 0 && Mo > 0 && No > 0 && target_workgroups >= 0, "urse_gemm_tn: bad argument");
  const long g_tn_target_wgs = target_workgroups > 0 ? target_workgroups : 256;
  const bool act_f16 = dtype == URSE_BF16_ACT_F16;       // A = bf16 gradients, B = the forward's f16 activations (converted in registers)
  if (act_f16) {
    if (!urse_gemm_tn_act_f16_supported(R, Mo, No, 0, colsum != nullptr, 1, 0) || perm_h != 0 || shift != 0 || period != 0) {
      set_error("urse_gemm_tn: URSE_BF16_ACT_F16 serves the shapes urse_gemm_tn_act_f16_supported accepts (R%ld Mo%ld No%ld)", (long)R, (long)Mo, (long)No);
      return URSE_ERR_UNSUPPORTED;
    }
    dtype = URSE_BF16;
  }
  const int es = dtype == URSE_BF16 ? 2 : 4;
  URSE_CHECK_ARG((lda * es) % 16 == 0 && (ldb * es) % 16 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                 "urse_gemm_tn: operands must be 16-byte aligned with 16-byte-multiple row pitch");
  URSE_CHECK_ARG(lda >= Mo && ldb >= No && ldc >= No, "urse_gemm_tn: leading dimension too small");
  TnArgs p;
  p.A = (const char*)A; p.B = (const char*)B; p.C = C; p.colsum = colsum;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.R = R; p.Mo = Mo; p.No = No;
  p.shift = shift; p.inner = inner > 0 ? inner : 1; p.period = period; p.invalid_step = invalid_step;
  p.perm_h = perm_h;
  p.B2 = nullptr; p.C2 = nullptr; p.ldb2 = p.ldc2 = p.No2 = p.nt1 = 0; p.pad_[0] = p.pad_[1] = 0;
  static const bool no_dma = getenv("URSE_TN_NO_DMA") != nullptr;
  if (dtype == URSE_BF16 && !no_dma && perm_h == 0 && shift == 0 && period == 0 && Mo < 512 && Mo >= 160 && No >= 512 &&
      R >= 16384 && R < (1L << 30)) {
    // wide-and-short gradient (fc weight [196, 784]): run the big kernel on the transposed problem
    TnArgs q = p;
    q.A = (const char*)B; q.lda = ldb; q.Mo = No;
    q.B = (const char*)A; q.ldb = lda; q.No = Mo;
    q.perm_h = TN_TRANSPOSED;
    const long pad7 = (q.No + 223) / 224 * 224, pad8 = (q.No + 255) / 256 * 256;
    const int ntw = pad7 < pad8 ? 7 : 8;
    const long bnx = 32L * ntw;
    const long tl = ((q.Mo + 255) / 256) * ((q.No + bnx - 1) / bnx);
    long slices = g_tn_target_wgs / tl;
    if (slices < 1) slices = 1;
    long rps = (R + slices - 1) / slices;
    rps = (rps + 31) / 32 * 32;
    slices = (R + rps - 1) / rps;
    q.rows_per_slice = rps;
    dim3 grid((unsigned)(tl * slices));
    note_launch(URSE_KV_TN_RING_T);
    if (act_f16) note_launch(URSE_KV_TN_ACT_F16);
    launch_tn_dma(ntw, colsum ? 2 : 0, grid, (hipStream_t)stream, q, act_f16 ? 1 : 0);
    URSE_CHECK_LAUNCH("urse_gemm_tn");
    return URSE_OK;
  }
  if (dtype == URSE_BF16 && !no_dma && Mo >= 512 && No >= 160 && R >= 16384 && inner < (1L << 31) && R < (1L << 30) && shift > -(1L << 30) && shift < (1L << 30) && period < (1L << 31)) {
    // big weight gradients: 256-wide tiles on the LDS-DMA ring, one workgroup per CU
    const long pad7 = (No + 223) / 224 * 224, pad8 = (No + 255) / 256 * 256;
    const int ntw = pad7 < pad8 ? 7 : 8;
    const long bnx = 32L * ntw;
    const long tl = ((Mo + 255) / 256) * ((No + bnx - 1) / bnx);
    long slices = g_tn_target_wgs / tl;
    if (slices < 1) slices = 1;
    long rps = (R + slices - 1) / slices;
    rps = (rps + 31) / 32 * 32;
    slices = (R + rps - 1) / rps;
    p.rows_per_slice = rps;
    dim3 grid((unsigned)(tl * slices));
    note_launch(URSE_KV_TN_RING);
    launch_tn_dma(ntw, colsum ? 1 : 0, grid, (hipStream_t)stream, p);
    URSE_CHECK_LAUNCH("urse_gemm_tn");
    return URSE_OK;
  }
  const long tiles = ((Mo + BM - 1) / BM) * ((No + BN - 1) / BN);
  const int bkr = dtype == URSE_BF16 ? 32 : 16;
  // one full round of resident workgroups (3 per CU at this kernel's register budget): a partial second round costs
  // up to 25 % on the big weight-gradient shapes
  long target = 768;
  if (const char* e = getenv("URSE_TN_TARGET")) target = atol(e);
  long slices = target / tiles;
  const long max_slices = (R + 4 * bkr - 1) / (4 * bkr);
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  long rps = (R + slices - 1) / slices;
  rps = (rps + bkr - 1) / bkr * bkr;
  slices = (R + rps - 1) / rps;
  p.rows_per_slice = rps;
  URSE_CHECK_ARG(tiles * slices < (1L << 31), "urse_gemm_tn: too many tiles");
  dim3 grid((unsigned)(tiles * slices));
  note_launch(URSE_KV_TN_128);
  if (dtype == URSE_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p);
  URSE_CHECK_LAUNCH("urse_gemm_tn");
  return URSE_OK;
}

extern "C" int urse_gemm_tn_grouped(const void* descs, int groups, int max_blocks, int dtype, void* stream) {
  URSE_CHECK_ARG(descs && groups > 0 && max_blocks > 0 && groups < 65536, "urse_gemm_tn_grouped: bad argument");
  dim3 grid((unsigned)max_blocks, (unsigned)groups);
  note_launch(URSE_KV_TN_GROUPED);
  if (dtype == URSE_BF16)
    hipLaunchKernelGGL(gemm_tn_grouped_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const TnArgs*)descs);
  else
    hipLaunchKernelGGL(gemm_tn_grouped_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const TnArgs*)descs);
  URSE_CHECK_LAUNCH("urse_gemm_tn_grouped");
  return URSE_OK;
}

// which weight-gradient shapes have a mixed-operand (URSE_BF16_ACT_F16) kernel: No2 == 0: urse_gemm_tn's transposed ring kernel with column sums
// (the fc gradient [196, 784]); No2 > 0: urse_gemm_tn_dual's 224 x 320 kernel (row / mask conditions are checked by the call itself)
extern "C" int urse_gemm_tn_act_f16_supported(int64_t R, int64_t Mo, int64_t No, int64_t No2, int with_colsum, int64_t inner, int64_t period) {
  static const bool no_dma = getenv("URSE_TN_NO_DMA") != nullptr, no224 = getenv("URSE_TN_NO_224") != nullptr;
  if (no_dma || R < 16384 || R >= (1L << 30)) return 0;
  if (No2 == 0) {
    const long pad7 = (Mo + 223) / 224 * 224, pad8 = (Mo + 255) / 256 * 256;        // (transposed: the kernel's column tiles run over Mo)
    return (with_colsum && Mo < 512 && Mo >= 160 && No >= 512 && pad7 < pad8) ? 1 : 0;
  }
  const long No8 = (No + 7) & ~7L, V = No8 + No2 + (with_colsum ? 1 : 0);
  if (inner < 1) inner = 1;
  // (the shifted operand's step mask must cover the rows the shift reaches: whole sequences of `period` steps, shift = -/+ inner)
  return (!no224 && Mo >= 512 && Mo % 224 == 0 && No2 % 8 == 0 && R % 32 == 0 && V <= 640 && period > 0 && period < (1L << 31) && inner < (1L << 31) &&
          R % (inner * period) == 0 && (Mo / 224) * ((V + 319) / 320) <= 256) ? 1 : 0;
}

// dW1[Mo, No] += A^T B (+ colsum),  dW2[Mo, No2] += A^T B2' (B2' = B2 shifted / masked as in urse_gemm_tn) in ONE pass
// over A: the two weight gradients of one LSTM direction (A = dgates [M, 4H], B = layer input, B2 = h_{t-1}).
extern "C" int urse_gemm_tn_dual(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                                 const void* B2, int64_t ldb2, float* C2, int64_t ldc2, int64_t R, int64_t Mo, int64_t No,
                                 int64_t No2, int64_t shift, int64_t inner, int64_t period, int64_t invalid_step,
                                 int64_t perm_h, int dtype, int target_workgroups, void* stream) {
  URSE_CHECK_ARG(A && B && C && B2 && C2 && R > 0 && Mo > 0 && No > 0 && No2 > 0 && target_workgroups >= 0,
                 "urse_gemm_tn_dual: bad argument");
  const long g_tn_target_wgs = target_workgroups > 0 ? target_workgroups : 256;
  static const bool no_dma = getenv("URSE_TN_NO_DMA") != nullptr;
  const bool act_f16 = dtype == URSE_BF16_ACT_F16;       // A = bf16 gradients, B / B2 = the forward's f16 activations
  if (act_f16) {
    if (!urse_gemm_tn_act_f16_supported(R, Mo, No, No2, colsum != nullptr, inner, period)) {
      set_error("urse_gemm_tn_dual: URSE_BF16_ACT_F16 serves the shapes urse_gemm_tn_act_f16_supported accepts (R%ld Mo%ld No%ld No2%ld)", (long)R, (long)Mo, (long)No, (long)No2);
      return URSE_ERR_UNSUPPORTED;
    }
    dtype = URSE_BF16;
  }
  const bool big = dtype == URSE_BF16 && !no_dma && Mo >= 512 && R >= 16384 && R < (1L << 30) && shift > -(1L << 30) && shift < (1L << 30) && period < (1L << 31) && inner < (1L << 31) &&
                   (lda * 2) % 16 == 0 && (ldb * 2) % 16 == 0 && (ldb2 * 2) % 16 == 0 && ((uintptr_t)A % 16) == 0 &&
                   ((uintptr_t)B % 16) == 0 && ((uintptr_t)B2 % 16) == 0;
  if (!big && act_f16) { set_error("urse_gemm_tn_dual: URSE_BF16_ACT_F16 needs 16-byte aligned operands"); return URSE_ERR_UNSUPPORTED; }
  if (!big) {
    int rc = urse_gemm_tn(A, lda, B, ldb, C, ldc, colsum, R, Mo, No, 0, 1, 0, 0, perm_h, dtype, target_workgroups, stream);
    if (rc) return rc;
    return urse_gemm_tn(A, lda, B2, ldb2, C2, ldc2, nullptr, R, Mo, No2, shift, inner, period, invalid_step, perm_h, dtype,
                        target_workgroups, stream);
  }
  URSE_CHECK_ARG(lda >= Mo && ldb >= No && ldc >= No && ldb2 >= No2 && ldc2 >= No2, "urse_gemm_tn_dual: leading dimension too small");
  TnArgs p;
  p.A = (const char*)A; p.B = (const char*)B; p.C = C; p.colsum = colsum;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.R = R; p.Mo = Mo; p.No = No;
  p.shift = shift; p.inner = inner > 0 ? inner : 1; p.period = period; p.invalid_step = invalid_step;
  p.perm_h = perm_h;
  p.B2 = (const char*)B2; p.C2 = C2; p.ldb2 = ldb2; p.ldc2 = ldc2; p.No2 = No2; p.pad_[0] = p.pad_[1] = 0;
  {
    // 224 x 320 tiles over the virtual operand [B | B2 | ones] when the rows are whole 224-row tiles and the columns fit two tiles
    static const bool no224 = getenv("URSE_TN_NO_224") != nullptr;
    const long No8 = (No + 7) & ~7L, V = No8 + No2 + (colsum ? 1 : 0);
    const bool mask_covers_range = period > 0 && R % ((inner > 0 ? inner : 1) * period) == 0 &&
                                   ((shift == -(inner > 0 ? inner : 1) && invalid_step == 0) || (shift == (inner > 0 ? inner : 1) && invalid_step == period - 1));
    if (!no224 && Mo % 224 == 0 && No2 % 8 == 0 && ldb2 >= No2 && R % 32 == 0 && mask_covers_range && V <= 640 && ldb >= No8 && perm_h >= 0 && lda >= Mo && ((Mo / 224) * ((V + 319) / 320)) <= g_tn_target_wgs) {
      const long tl = (Mo / 224) * ((V + 319) / 320);
      long slices = g_tn_target_wgs / tl;
      if (slices < 1) slices = 1;
      long rps = (R + slices - 1) / slices;
      rps = (rps + 31) / 32 * 32;
      slices = (R + rps - 1) / rps;
      p.rows_per_slice = rps;
      p.nt1 = 0;
      {
        // A/B switch, read per call: 2 = two stages per barrier with the queue drained (default), 3 = three stages in flight and one barrier per stage.
        // Alone the deeper form is 2 % faster (profiles/r05_exp_tn224_depth_v1.log); in the train step it LOSES 2.1 ms, same box, both orders
        // (130.38 / 130.73 against 132.91 / 132.47, profiles/r05_ab_tn224_depth_v3.log).  An earlier A/B had said the opposite - on a binary that held
        // both loops in one kernel behind a run-time branch and spilled nine registers in either (see the template note at the kernel).
        const char* e = getenv("URSE_TN224_DEPTH");
        p.pad_[0] = e ? atol(e) : 2;
      }
      note_launch(URSE_KV_TN_DUAL);
      if (act_f16) note_launch(URSE_KV_TN_ACT_F16);
      if (act_f16) hipLaunchKernelGGL((gemm_tn_dual224_kernel<2, true>), dim3((unsigned)(tl * slices)), dim3(512), 0, (hipStream_t)stream, p);
#ifdef URSE_EXPERIMENTS      // (DMA issue between the MFMA groups, round 6: -8.5 % alone, neutral in the step, profiles/r06_ab_tn224_interleave_v1 / _v2.log - variant builds only)
      else if (p.pad_[0] == 4) hipLaunchKernelGGL(gemm_tn_dual224_kernel<4>, dim3((unsigned)(tl * slices)), dim3(512), 0, (hipStream_t)stream, p);
#endif
#ifdef URSE_EXPERIMENTS      // (three stages in flight: +2 % alone, -2.1 ms LOST in the step, round 5 - variant builds only)
      else if (p.pad_[0] == 3) hipLaunchKernelGGL(gemm_tn_dual224_kernel<3>, dim3((unsigned)(tl * slices)), dim3(512), 0, (hipStream_t)stream, p);
#endif
      else hipLaunchKernelGGL(gemm_tn_dual224_kernel<2>, dim3((unsigned)(tl * slices)), dim3(512), 0, (hipStream_t)stream, p);
      URSE_CHECK_LAUNCH("urse_gemm_tn_dual");
      return URSE_OK;
    }
  }
  if (act_f16) { set_error("urse_gemm_tn_dual: URSE_BF16_ACT_F16: rows / step mask outside what the 224 x 320 kernel takes"); return URSE_ERR_UNSUPPORTED; }
  const long bnx = 224;                                   // 7 column tiles per wave: 196 -> 224, 392 -> 448
  p.nt1 = (No + bnx - 1) / bnx;
  const long tl = ((Mo + 255) / 256) * (p.nt1 + (No2 + bnx - 1) / bnx);
  long slices = g_tn_target_wgs / tl;
  if (slices < 1) slices = 1;
  long rps = (R + slices - 1) / slices;
  rps = (rps + 31) / 32 * 32;
  slices = (R + rps - 1) / rps;
  p.rows_per_slice = rps;
  note_launch(URSE_KV_TN_DUAL);
  launch_tn_dma(7, colsum ? 1 : 0, dim3((unsigned)(tl * slices)), (hipStream_t)stream, p);
  URSE_CHECK_LAUNCH("urse_gemm_tn_dual");
  return URSE_OK;
}

#ifdef T224STAMP
extern "C" int urse_diag_tn224_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_t224stamps), sizeof(unsigned long long) * 512 * 8);
}
#endif
