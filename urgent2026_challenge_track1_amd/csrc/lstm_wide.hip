// "Wide" streaming LSTM recurrence (bf16): 64 sequences per workgroup, so that every recurrent-weight byte streamed
// from L2 feeds four MFMA row tiles instead of one.
//
// lstm.hip gives a workgroup 16 sequences; at the C2 band path (12,832 sequences x 34 steps) it is bound by the
// per-CU L1 rate at which W_hh (1.3 MB per direction) is re-streamed every step.  Here a workgroup of 8 waves owns
// 64 sequences of one direction for the whole time loop:
//   * h_{t-1} [64, Hp] bf16 sits in LDS (MFMA A operand), double-buffered so a step needs ONE barrier;
//   * the hidden units are cut into blocks of 16; a wave owns 3-4 blocks and for each block ("group") accumulates
//     the four gates x four row tiles = 16 MFMA tiles, K = Hp, with the weights read as B fragments (1 KiB per
//     wave-instruction, consumption order == memory order) through a 13-deep register ring that runs ahead across
//     groups and steps;
//   * because a group's four column tiles are the four GATES of the same 16 units, i/f/g/o of one (row, unit) land in
//     the same lane and accumulator slot: the cell update needs no cross-lane traffic;
//   * the accumulators start from the gate pre-activations (prefetched one group ahead), c_{t-1} is re-read from the
//     f32 cell-state stream the same lane wrote one step earlier (16 registers per block instead of 64 resident);
//   * h_t goes to the other LDS buffer and from there to HBM with 16-byte stores after the barrier.
// Same math, layouts and outputs as lstm.hip (gate-interleaved gx overwritten by the activations, f32 c, bf16 h);
// replaces the cuDNN LSTM under espnet2 BSRNN's rnn_freq (reference twin baseline_code/models/bsrnn_flowse.py:303-306).
#include <stdlib.h>

#include "urse_common.h"
#include "fft_lds.h"   // fastdiv

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// Geometry (template parameters): RT row tiles (16 * RT sequences) and WW waves per workgroup.
//   <4, 8>: 64 sequences, one workgroup per CU -- the least weight traffic per sequence;
//   <2, 4>: 32 sequences, TWO workgroups per CU (54 KB of LDS, 4 waves x 256 VGPRs each) whose phases drift apart, so that
//           one's cell math / stores overlap the other's weight stream and MFMAs.

struct WideArgs {
  void* gx; long ldg;
  const void* whhb;               // [2][nblk][NSLAB][4 gates][64 lanes][16 B]
  void* hout; long ldh;
  float* c;
  int H, Hp, save;
  long inner, outer, stride;
  int n_seq, seq_len;
  unsigned m_cpr;                 // fastdiv magic of the 16-B chunks per h row
  int xcd;                        // xcd_dir_tile mapping
};

template <int NSLAB, int MAXG, int RT, int WW>
__global__ void __launch_bounds__(WW * 64, 2) lstm_fwd_wide_kernel(WideArgs p) {
  constexpr int WTHR = WW * 64, WROWS = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps block bookkeeping and weight bases in SGPRs
  int dir, tile_;
  xcd_dir_tile(p.xcd, dir, tile_);
  const int H = p.H;
  constexpr int Hp = NSLAB * 32, pitch = lds_frag_pitch(Hp * 2);      // compile-time: LDS offsets of the k loop fold into immediates
  const int seq0 = tile_ * WROWS;
  const int nrows = min(WROWS, p.n_seq - seq0);
  int* rowtab = reinterpret_cast<int*>(smem + 2 * WROWS * pitch);   // row index of (sequence, t = 0)

  const int nblk = (H + 15) >> 4;
  const int base = nblk / WW, rem = nblk % WW;
  const int nmy = base + (w < rem ? 1 : 0);
  const int b0 = w * base + min(w, rem);

  for (int i = tid; i < 2 * WROWS * pitch / 16; i += WTHR) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (tid < WROWS) {
    int seq = seq0 + tid;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    rowtab[tid] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }

  constexpr int NF = 4 * NSLAB;                  // fragments per block
  constexpr long BLKB = (long)NF * 1024;         // bytes per block
  const char* wbase = reinterpret_cast<const char*>(p.whhb) + (long)dir * nblk * BLKB;   // scalar base, lane offset below
  const unsigned loff = lane * 16;
  uint4 ring[NSLAB];
  if (nmy > 0) {
#pragma unroll
    for (int f = 0; f < NSLAB; ++f) ring[f] = *reinterpret_cast<const uint4*>(wbase + (long)b0 * BLKB + f * 1024 + loff);
  }
  bf16_t* gx = reinterpret_cast<bf16_t*>(p.gx);
  bf16_t* hout = reinterpret_cast<bf16_t*>(p.hout);
  const int cpr = (H * 2 + 15) / 16;
  const bool hvec = ((((long)dir * H * 2) | (p.ldh * 2)) & 15) == 0 && ((reinterpret_cast<uintptr_t>(p.hout) & 15) == 0);
  __syncthreads();

  uint2 gxn[RT][4];
  // row indices and leading dimensions are 32-bit (M < 2^31 rows is checked on the host): an address is then one
  // v_mad_i64_i32 instead of a 64 x 64-bit multiply (address arithmetic was a third of this kernel's VALU work)
  const int ldg_i = (int)p.ldg, ldc_i = 2 * H, ldh_i = (int)p.ldh, gcol_i = dir * 4 * H, stride_i = (int)p.stride;
  auto load_gx = [&](int blk, int toff_) {
    int uu = blk * 16 + lc;
    if (uu >= H) uu = H - 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowtab[rt * 16 + lr * 4 + r] + toff_;
        gxn[rt][r] = *reinterpret_cast<const uint2*>(gx + ((long)row * ldg_i + (gcol_i + uu * 4)));
      }
  };
  if (nmy > 0) load_gx(b0, (dir ? p.seq_len - 1 : 0) * stride_i);
  float cnx[RT][4];                               // c_{t-1} of the next group (prefetched with its pre-activations)
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) cnx[a][b] = 0.f;
  auto load_c = [&](int blk, int toff_) {
    int uu = blk * 16 + lc;
    if (uu >= H) uu = H - 1;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        cnx[rt][r] = p.c[(long)(rowtab[rt * 16 + lr * 4 + r] + toff_) * ldc_i + (dir * H + uu)];
  };

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const int toff_prev = (dir ? t + 1 : t - 1) * stride_i;
    const int toff_next = (dir ? t - 1 : t + 1) * stride_i;
    const char* hc = smem + (step & 1) * WROWS * pitch;
    char* hn = smem + ((step & 1) ^ 1) * WROWS * pitch;

#pragma unroll 1
    for (int g = 0; g < MAXG; ++g) {
      if (g < nmy) {
        const int gn = (g + 1 < nmy) ? g + 1 : 0;
        const char* wcur = wbase + (long)(b0 + g) * BLKB;
        const char* wnext = wbase + (long)(b0 + gn) * BLKB;
        const int u = (b0 + g) * 16 + lc;
        const bool uvalid = u < H;
        // acc starts from the gate pre-activations x*W_ih + b (prefetched during the previous group's cell phase):
        // 8 B per (row, unit), 16 lanes cover one 128-byte line
        f32x4_t acc[4][RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint2 gv2 = gxn[rt][r];
            acc[0][rt][r] = __uint_as_float(gv2.x << 16);
            acc[1][rt][r] = __uint_as_float(gv2.x & 0xffff0000u);
            acc[2][rt][r] = __uint_as_float(gv2.y << 16);
            acc[3][rt][r] = __uint_as_float(gv2.y & 0xffff0000u);
          }
        // c_{t-1} of this block comes back from the cell-state stream this lane wrote one step ago (16 registers per
        // block instead of 64 resident ones)
        float cprev[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cprev[rt][r] = cnx[rt][r];
        uint4 an[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const uint4*>(hc + (rt * 16 + lc) * pitch + 16 * lr);
#pragma unroll
        for (int ks = 0; ks < NSLAB; ++ks) {
          uint4 a[RT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) a[rt] = an[rt];
          if (ks + 1 < NSLAB) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              an[rt] = *reinterpret_cast<const uint4*>(hc + (rt * 16 + lc) * pitch + (ks + 1) * 64 + 16 * lr);
          }
#pragma unroll
          for (int gate = 0; gate < 4; ++gate) {
            const int f = ks * 4 + gate, slot = f % NSLAB, f2 = f + NSLAB;
            const uint4 b = ring[slot];
#ifndef ABL_NO_MFMA
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[gate][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[rt]),
                                                                      __builtin_bit_cast(bf16x8_t, b), acc[gate][rt], 0, 0, 0);
#else
            acc[gate][0][0] += __uint_as_float(b.x ^ a[0].x);
#endif
#ifndef ABL_NO_W
            ring[slot] = (f2 < NF) ? *reinterpret_cast<const uint4*>(wcur + f2 * 1024 + loff)
                                   : *reinterpret_cast<const uint4*>(wnext + (f2 - NF) * 1024 + loff);
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // pre-activations of the next group (next step's first group after the last one)
        {
          const bool wrap = g + 1 >= nmy;
#ifndef ABL_NO_LD
          if (!wrap || step + 1 < p.seq_len) load_gx(b0 + gn, wrap ? toff_next : toff);
          // c_{t-1} of the next group: previous step's rows, or (after the last group) the rows this lane wrote during
          // THIS step for its first block (group 0 != this group unless the wave owns one block, handled below)
          if (!wrap) {
            if (step > 0) load_c(b0 + gn, toff_prev);
          } else if (nmy > 1 && step + 1 < p.seq_len) {
            load_c(b0, toff);
          }
#endif
        }
        // cell update: acc[gate][rt][r] = gate `gate` of (row rt*16 + lr*4 + r, unit u)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gi = acc[0][rt][r], gf = acc[1][rt][r], gg = acc[2][rt][r], go = acc[3][rt][r];
#ifdef ABL_NO_CELL
            const float iv = gi, fv = gf, gv = gg, ov = go;
            const float cv = fv * cprev[rt][r] + iv * gv;
            const float hv = ov * cv;
#else
            const float iv = sigmoidf_(gi), fv = sigmoidf_(gf), gv = tanhf_(gg), ov = sigmoidf_(go);
            // one contraction, spelled out: left to the compiler, 15 of the 16 unrolled instances became mul + fma and one a packed
            // multiply + add (1 ulp apart), which made bit comparisons with lstm_rw.hip impossible
            const float cv = __builtin_fmaf(fv, cprev[rt][r], __fmul_rn(iv, gv));
            const float hv = ov * tanhf_(cv);
#endif
            if (nmy == 1) cnx[rt][r] = cv;       // a wave with one block keeps its cell state in registers
            const int lrow = rt * 16 + lr * 4 + r;
            if (uvalid) {
              *reinterpret_cast<bf16_t*>(hn + lrow * pitch + u * 2) = f32_to_bf16(hv);
#ifndef ABL_NO_ST
              if (lrow < nrows) {
                const int row = rowtab[lrow] + toff;
                p.c[(long)row * ldc_i + (dir * H + u)] = cv;
                if (p.save) {
                  uint2 sv;
                  sv.x = (unsigned)f32_to_bf16(iv) | ((unsigned)f32_to_bf16(fv) << 16);
                  sv.y = (unsigned)f32_to_bf16(gv) | ((unsigned)f32_to_bf16(ov) << 16);
                  *reinterpret_cast<uint2*>(gx + ((long)row * ldg_i + (gcol_i + u * 4))) = sv;
                }
              }
#endif
            }
          }
      }
    }
    __syncthreads();
    // h_t -> hout, 16-byte chunks of full rows
    for (int idx = tid; idx < WROWS * cpr; idx += WTHR) {
      const int row = fastdiv(idx, p.m_cpr), cc = idx - row * cpr;
      if (row >= nrows) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(hn + row * pitch + cc * 16);
      const int grow = rowtab[row] + toff;
      bf16_t* dst = hout + ((long)grow * ldh_i + (dir * H + cc * 8));
      if (hvec && cc * 8 + 8 <= H) {
        *reinterpret_cast<uint4*>(dst) = v;
      } else {
        const bf16_t* sv = reinterpret_cast<const bf16_t*>(&v);
        for (int e = 0; e < 8 && cc * 8 + e < H; ++e) dst[e] = sv[e];
      }
    }
  }
}

// block-ordered recurrent weights: (dir, blk, ks, gate) = 64 lanes x 16 B; lane (lr, lc): unit blk*16 + lc,
// k = ks*32 + 8*lr + j
__device__ __forceinline__ void lstm_pack_blocks_dev(const float* __restrict__ whh, bf16_t* __restrict__ out, int H, int Hp) {
  const int nblk = (H + 15) >> 4, nslab = Hp / 32, G4 = 4 * H;
  const long total = (long)2 * nblk * nslab * 4 * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int jj = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int g = (int)(r % 4); r /= 4;
    const int ks = (int)(r % nslab); r /= nslab;
    const int blk = (int)(r % nblk);
    const int d = (int)(r / nblk);
    const int lc = lane & 15, lr = lane >> 4;
    const int u = blk * 16 + lc, k = ks * 32 + 8 * lr + jj;
    out[idx] = f32_to_bf16((u < H && k < H) ? whh[((long)d * G4 + g * H + u) * H + k] : 0.f);
  }
}
__global__ void __launch_bounds__(256) lstm_pack_blocks_kernel(const float* __restrict__ whh, bf16_t* __restrict__ out, int H, int Hp) {
  lstm_pack_blocks_dev(whh, out, H, Hp);
}
__global__ void __launch_bounds__(256) lstm_pack_blocks_multi_kernel(const PackRow* __restrict__ tab, int H, int Hp) {
  const PackRow r = tab[blockIdx.y];
  if (r.whhb) lstm_pack_blocks_dev(r.whh, (bf16_t*)r.whhb, H, Hp);
}

template <int NSLAB, int MAXG, int RT, int WW>
static int launch_wide_fwd(const WideArgs& p, hipStream_t st) {
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_wide_kernel<NSLAB, MAXG, RT, WW>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  constexpr int WROWS = 16 * RT;
  const size_t lds = (size_t)2 * WROWS * lds_frag_pitch(p.Hp * 2) + WROWS * sizeof(int);
  dim3 grid((p.n_seq + WROWS - 1) / WROWS, 2);
  hipLaunchKernelGGL((lstm_fwd_wide_kernel<NSLAB, MAXG, RT, WW>), grid, dim3(WW * 64), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_wide_fwd");
  return URSE_OK;
}

static bool wide_shape(int H, int Hp, int* nslab, int* maxg) {
  if (H <= 0 || Hp % 32 || Hp < H) return false;
  const int nblk = (H + 15) / 16;
  *nslab = Hp / 32;
  *maxg = (nblk + 8 - 1) / 8;
  return (*nslab == 13 && *maxg == 4) || (*maxg == 1 && *nslab >= 1 && *nslab <= 4);
}

}  // namespace urse

using namespace urse;

extern "C" int urse_lstm_wide_supported(int H, int Hp) {
  int a, b;
  return wide_shape(H, Hp, &a, &b) ? 1 : 0;
}

extern "C" int urse_lstm_pack_blocks(const float* whh, void* out, int H, int Hp, void* stream) {
  URSE_CHECK_ARG(whh && out && H > 0 && Hp % 32 == 0 && Hp >= H, "urse_lstm_pack_blocks: bad argument");
  hipLaunchKernelGGL(lstm_pack_blocks_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, whh, (bf16_t*)out, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_blocks");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_blocks_multi(const void* table, int n_lstm, int H, int Hp, void* stream) {
  URSE_CHECK_ARG(table && n_lstm > 0 && n_lstm < 65536 && H > 0 && Hp % 32 == 0 && Hp >= H, "urse_lstm_pack_blocks_multi: bad argument");
  hipLaunchKernelGGL(lstm_pack_blocks_multi_kernel, dim3(256, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_blocks_multi");
  return URSE_OK;
}

extern "C" int urse_lstm_wide_fwd(void* gx, int64_t ldg, const void* whhb, void* hout, int64_t ldh, float* c, int H, int Hp,
                                  int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save,
                                  void* stream) {
  URSE_CHECK_ARG(gx && whhb && hout && c, "urse_lstm_wide_fwd: null pointer (c is required: it carries c_{t-1})");
  int nslab, maxg;
  URSE_CHECK_ARG(wide_shape(H, Hp, &nslab, &maxg), "urse_lstm_wide_fwd: unsupported H=%d Hp=%d", H, Hp);
  URSE_CHECK_ARG(n_seq > 0 && seq_len > 0 && inner > 0, "urse_lstm_wide_fwd: bad sequence geometry");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldh < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_wide_fwd: row indices must fit 32 bits");
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 4 == 0 && ldh >= 2L * H && ((uintptr_t)gx % 8) == 0,
                 "urse_lstm_wide_fwd: bad leading dimension / alignment");
  WideArgs p;
  p.gx = gx; p.ldg = ldg; p.whhb = whhb; p.hout = hout; p.ldh = ldh; p.c = c; p.H = H; p.Hp = Hp; p.save = save;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  p.m_cpr = fastdiv_magic((unsigned)((H * 2 + 15) / 16));
  p.xcd = (xcd_dir_env() >> 1) & 1;
  hipStream_t st = (hipStream_t)stream;
  static const int variant = getenv("URSE_WIDE_VARIANT") ? atoi(getenv("URSE_WIDE_VARIANT")) : 0;
  note_launch(URSE_KV_LSTM_FWD_WIDE);
  if (nslab == 13) {
    if (variant == 1) return launch_wide_fwd<13, 7, 2, 4>(p, st);      // 32 sequences, two workgroups per CU
    return launch_wide_fwd<13, 4, 4, 8>(p, st);
  }
  switch (nslab) {
    case 1: return launch_wide_fwd<1, 1, 4, 8>(p, st);
    case 2: return launch_wide_fwd<2, 1, 4, 8>(p, st);
    case 3: return launch_wide_fwd<3, 1, 4, 8>(p, st);
    default: return launch_wide_fwd<4, 1, 4, 8>(p, st);
  }
}
