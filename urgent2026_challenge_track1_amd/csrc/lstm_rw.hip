// "Row-wave" LSTM recurrence for many short sequences (bf16): the band path of BSRNN (12,832 sequences x 34 steps per
// direction at C2; espnet2 BSRNN's rnn_freq, reference twin baseline_code/models/bsrnn_flowse.py:303-306).
//
// The streaming kernels (lstm.hip, lstm_wide.hip) give a workgroup 16-64 sequences and make every wave stream its own
// share of W_hh (1.3 MB per direction) from L2 into registers on every step; the phases of a step (weight stream + MFMA,
// cell math, stores) then run one after the other behind a workgroup barrier, and 401 + 401 workgroups of 64 sequences
// fill 1.57 rounds of the 256 CUs.  Here the division of labour is turned around:
//   * a WAVE owns 16 sequences for the whole time loop.  h_{t-1} of its sequences is REGISTER resident as the MFMA A
//     operand (13 k-slabs x 4 VGPRs for Hp = 416); nothing about a sequence is shared between waves, so there is no
//     workgroup barrier on the data path and no inter-workgroup hand-off;
//   * the seven compute waves of a workgroup share ONE pass over W_hh per step: an eighth wave (the loader) streams the
//     block-ordered fragments L2 -> LDS with LDS-DMA (global_load_lds_dwordx4, no staging registers) through a ring of
//     five 13 KB slots, three stages in flight; a compute wave reads a fragment from LDS (ds_read_b128, 1 KiB, lane
//     linear = conflict free) four MFMAs ahead of its use.  112 sequences per weight pass instead of 64, and the
//     weights travel L2 -> CU once per 112 rows;
//   * a stage (13 fragments = one quarter of a 16-unit block) ends with a raw s_barrier: it tells the loader that the
//     slot before it is free and the compute waves that the stage after the next has landed (counted vmcnt on the
//     loader's side).  Waves that own no tile exit at once (a barrier counts live waves only);
//   * the cell update of block b - 1 (sigmoid / tanh, c_t, h_t, stores) is issued quarter by quarter between the MFMAs of
//     block b, so that the SIMD's matrix pipe and its vector ALU work at the same time; the gate pre-activations and
//     c_{t-1} of a block are fetched three blocks (~3.5 us) ahead of their use;
//   * h_t goes, 2 bytes per (sequence, unit), to a wave-private [16][Hp] LDS tile; at the end of the step the wave
//     reads it back as next step's A fragments and writes the rows to hout with 16-byte stores.
// The grid is 2 x G workgroups, G chosen so that the tiles of a direction are dealt 6-7 per workgroup over half the
// chip (802 tiles -> 34 workgroups of 7 + 94 of 6); direction = blockIdx.x & 1, so that under the round-robin dealing
// of workgroups to XCDs every XCD's L2 serves ONE direction's weights (speed only).
// Same math, layouts and outputs as lstm_wide.hip, bit for bit (accumulators start from the pre-activations, k-slabs
// in ascending order, same cell functions): gate-interleaved gx overwritten by the activations, f32 c, bf16 h.
#include <stdlib.h>

#include <type_traits>

#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int RW_MAXT = 7;        // compute waves (16-sequence tiles) per workgroup
constexpr int RW_WAVES = 8;       // + the loader
constexpr int RW_NSLOT = 5;       // ring slots
constexpr int RW_UB = 5;          // blocks per unrolled body (ring slot and prefetch slot indices become constants)
constexpr int RW_D = 4;           // weight fragments read ahead of their MFMA
constexpr int RW_PD = 3;          // blocks the gate pre-activations / c_{t-1} are fetched ahead

struct RwArgs {
  void* gx; long ldg;
  const void* whhb;               // [2][NBLK][NSLAB][4 gates][64 lanes][16 B]   (urse_lstm_pack_blocks)
  void* hout; long ldh;
  float* c;
  int save;
  long inner, outer, stride;
  int n_seq, seq_len;
  unsigned gx_bytes, c_bytes, h_bytes;     // buffer sizes (range-checked accesses)
  int tiles_base, tiles_rem;      // workgroup i of a direction owns tiles [i * base + min(i, rem), + base + (i < rem))
};

// timing diagnostics (wrong results): RWABL_NO_DMA, RWABL_NO_DMA_WAIT, RWABL_NO_BARRIER, RWABL_NO_MFMA, RWABL_NO_CELL, RWABL_NO_LOAD,
// RWABL_NO_STORE, RWABL_NO_HOUT (scripts/abl_rw.py)
#ifdef RWABL_NO_BARRIER
#define RW_BARRIER() do { } while (0)
#else
#define RW_BARRIER() __builtin_amdgcn_s_barrier()
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS byte address `dst` + lane * 16
__device__ __forceinline__ void rw_glds16(const char* gsrc, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int H, int HP, bool SAVE>
__global__ void __launch_bounds__(RW_WAVES * 64, 2) lstm_fwd_rw_kernel(RwArgs p) {
  constexpr int NBLK = (H + 15) / 16, NSLAB = HP / 32, NF = 4 * NSLAB;   // 25 blocks of 16 units, 13 k-slabs, 52 fragments per block
  constexpr int SF = NSLAB;                                             // fragments per stage (a quarter block)
  constexpr int SPS = NBLK * 4;                                         // stages per step
  constexpr int SLOTB = SF * 1024;
  constexpr int HPITCH = lds_frag_pitch(HP * 2);
  static_assert(NBLK % RW_UB == 0 && (RW_UB * 4) % RW_NSLOT == 0, "ring / prefetch slots must be static in the unrolled body");
  static_assert((RW_UB * NF) % RW_D == 0, "fragment read-ahead ring must close over the unrolled body");
  static_assert(SF == 13, "the loader's counted vmcnt is written for 13 DMAs per stage");
  static_assert(H % 8 == 0, "whole 16-byte chunks per h row");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                                    // [NSLOT][SF][1 KiB]
  char* hsb = smem + RW_NSLOT * SLOTB;                                  // [MAXT][16][HPITCH]

  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x & 1, wi = blockIdx.x >> 1;
  const int ntl = p.tiles_base + (wi < p.tiles_rem ? 1 : 0);
  const int tile0 = wi * p.tiles_base + min(wi, p.tiles_rem);
  const long total_stages = (long)p.seq_len * SPS;

  if (w == RW_WAVES - 1) {
    // ---------------- loader: stage s (13 KB) -> slot s % 5, three stages in flight ----------------
    // Invariant at barrier A_k: stages <= k + 1 have landed, the compute waves are done with stages <= k - 1.
    const char* wsrc = reinterpret_cast<const char*>(p.whhb) + (long)dir * NBLK * NF * 1024 + lane * 16;
    const unsigned ring0 = (unsigned)(size_t)ring;
    int sm = 0, slot = 0;                                               // stage within the step, ring slot of the next stage to issue
    auto issue = [&]() {
      const char* src = wsrc + (long)sm * SLOTB;
      const unsigned dst = ring0 + slot * SLOTB;
#ifndef RWABL_NO_DMA
#pragma unroll
      for (int f = 0; f < SF; ++f) rw_glds16(src + f * 1024, dst + f * 1024);
#endif
      sm = (sm + 1 == SPS) ? 0 : sm + 1;
      slot = (slot + 1 == RW_NSLOT) ? 0 : slot + 1;
    };
    issue(); issue(); issue(); issue();
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");                   // stages 0, 1 landed
    issue();
    RW_BARRIER();                                                       // A_0
    for (long k = 0; k < total_stages; ++k) {
      if (k >= 1) issue();                                              // stage k + 4 into the slot stage k - 1 has left
#ifndef RWABL_NO_DMA_WAIT
      asm volatile("s_waitcnt vmcnt(26)" ::: "memory");                 // stage k + 2 landed
#endif
      RW_BARRIER();                                                     // A_{k+1}
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the run-ahead stages must not outlive the workgroup
    return;
  }
  if (w >= ntl) return;                                                 // no tile: the barriers count live waves only

  // ---------------- compute wave: 16 sequences ----------------
  // Global accesses go through buffer resources with 32-bit byte offsets (host-checked: every matrix < 4 GB): the per-lane part
  // of an address is loop invariant, the step / block part is a scalar offset, and a lane that must not store (row past n_seq,
  // unit past H) gets an offset outside the buffer, which the hardware drops - the cell update has no branch and is scheduled
  // between the MFMAs.
  char* hs = hsb + w * 16 * HPITCH;
  const int seq0 = (tile0 + w) * 16;
  for (int i = lane; i < 16 * HPITCH / 16; i += 64) reinterpret_cast<uint4*>(hs)[i] = make_uint4(0, 0, 0, 0);
  constexpr unsigned OOB = 0xFFFFF000u;
  const int ldg_i = (int)p.ldg, ldc_i = 2 * H, ldh_i = (int)p.ldh, gcol_i = dir * 4 * H, hcol_i = dir * H, stride_i = (int)p.stride;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gx, 0, (int)p.gx_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  unsigned goff[4], coff[4], gsto[4], csto[4], hsoff[4];                // per-lane byte offsets of (row 4 lr + r, unit lc) at t = 0, block 0
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int seq = seq0 + lr * 4 + r;
    const bool rvalid = seq < p.n_seq;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const unsigned row = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    goff[r] = (row * (unsigned)ldg_i + (unsigned)(gcol_i + lc * 4)) * 2u;
    coff[r] = (row * (unsigned)ldc_i + (unsigned)(hcol_i + lc)) * 4u;
    gsto[r] = rvalid ? goff[r] : OOB;
    csto[r] = rvalid ? coff[r] : OOB;
    hsoff[r] = (unsigned)((lr * 4 + r) * HPITCH + lc * 2);
  }
  // hout rows of this wave's tile: chunk idx = lane + 64 k (k < 13) is (row idx / 49, 16-byte chunk idx % 49)
  constexpr int CPR = H * 2 / 16, HK = (16 * CPR + 63) / 64;            // 49 chunks per row, 13 chunks per lane
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  unsigned hoff[HK], hlds[HK];
#pragma unroll
  for (int k = 0; k < HK; ++k) {
    const int idx = lane + 64 * k, row = idx / CPR, cc = idx - row * CPR;
    const bool ok = idx < 16 * CPR && seq0 + row < p.n_seq;
    int seq = seq0 + row;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const unsigned grow = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    hoff[k] = ok ? (grow * (unsigned)ldh_i + (unsigned)(hcol_i + cc * 8)) * 2u : OOB;
    hlds[k] = (unsigned)((idx < 16 * CPR ? row : 0) * HPITCH + cc * 16);
  }

  uint4 hfrag[NSLAB];                                                   // h_{t-1}: A fragments (row lc, k = 32 ks + 8 lr ..)
#pragma unroll
  for (int ks = 0; ks < NSLAB; ++ks) hfrag[ks] = make_uint4(0, 0, 0, 0);
  uint2 gxq[RW_UB][4];                                                  // gate pre-activations, slot = block % 5 (three or four live)
  float cq[RW_UB][4];                                                   // c_{t-1}
#pragma unroll
  for (int a = 0; a < RW_UB; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gxq[a][r] = make_uint2(0u, 0u); cq[a][r] = 0.f; }
  // block `blk` of the step whose rows sit `toff_` rows after t = 0; c_{t-1} from `toffc_` (has_c false: the offset leaves the
  // buffer and the hardware returns zeros - no select behind the load, which the scheduler would wait for at once)
  auto prefetch = [&](int slot, int blk, int toff_, int toffc_, bool has_c) {
    const int sg = (toff_ * ldg_i + blk * 64) * 2, sc = (toffc_ * ldc_i + blk * 16) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifdef RWABL_NO_LOAD
      const u32x2 g2 = u32x2{(unsigned)sg, goff[r]};
      const float cv = __uint_as_float((unsigned)sc & 0x3fffffffu);
#else
#ifdef RWABL_NO_GLOAD
      const u32x2 g2 = u32x2{(unsigned)sg, goff[r]};
#else
      const u32x2 g2 = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)goff[r], sg, 0);
#endif
#ifdef RWABL_NO_CLOAD
      const float cv = __uint_as_float((unsigned)sc & 0x3fffffffu);
#else
      const float cv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(has_c ? coff[r] : OOB), sc, 0));
#endif
#endif
      gxq[slot][r] = make_uint2(g2[0], g2[1]);
      cq[slot][r] = cv;
    }
  };
  {
    const int t0 = dir ? p.seq_len - 1 : 0;
#pragma unroll
    for (int b = 0; b < RW_PD; ++b) prefetch(b, b, t0 * stride_i, t0 * stride_i, false);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the zeroed tile is this wave's own LDS write
  RW_BARRIER();                                                         // A_0: stages 0 and 1 are in the ring
  const unsigned loff = lane * 16;
  uint4 bq[RW_D];
#pragma unroll
  for (int i = 0; i < RW_D; ++i) bq[i] = *reinterpret_cast<const uint4*>(ring + i * 1024 + loff);

  f32x4_t accp[4];                                                      // gates of the previous block, waiting for their cell update
  float cprevp[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) accp[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) cprevp[r] = 0.f;
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto pack2 = [](float a, float b) -> unsigned {                       // one v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  };

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const bool more = step + 1 < p.seq_len;
    const int toff_next = more ? (dir ? t - 1 : t + 1) * stride_i : toff;      // (past the last step: harmless re-fetch of this one)
    const int toff_prev = step > 0 ? (dir ? t + 1 : t - 1) * stride_i : toff;
    const int sg_t = toff * ldg_i * 2, sc_t = toff * ldc_i * 4;
    const int sh_prev = toff_prev * ldh_i * 2;
    const unsigned hmask = step > 0 ? 0u : OOB;                          // (nothing to write out in front of the first step)

    // cell update of row 4 lr + r of block `blk` from accp / cprevp; TAIL: the block may hold units past H (the last one)
    auto cell = [&](int blk, int r, auto tail) {
#ifdef RWABL_CHEAP_CELL
      const float iv = accp[0][r], fv = accp[1][r], gv = accp[2][r], ov = accp[3][r];
      const float cv = __builtin_fmaf(fv, cprevp[r], __fmul_rn(iv, gv));
      const float hv = ov * cv;
#else
      const float iv = sigmoidf_(accp[0][r]), fv = sigmoidf_(accp[1][r]), gv = tanhf_(accp[2][r]), ov = sigmoidf_(accp[3][r]);
      const float cv = __builtin_fmaf(fv, cprevp[r], __fmul_rn(iv, gv));       // the contraction lstm_wide.hip spells out too
      const float hv = ov * tanhf_(cv);
#endif
      // units past H land in the tile's k padding (finite values against zero weights)
      *reinterpret_cast<bf16_t*>(hs + hsoff[r] + blk * 32) = f32_to_bf16(hv);
      unsigned co = csto[r], go = gsto[r];
      if constexpr (decltype(tail)::value) {
        const bool uvalid = blk * 16 + lc < H;
        co = uvalid ? co : OOB;
        go = uvalid ? go : OOB;
      }
#ifdef RWABL_NO_STORE
      co = hv == 123.f ? co : OOB; go = hv == 123.f ? go : OOB;
#endif
#ifdef RWABL_NO_CSTORE
      co = hv == 123.f ? co : OOB;
#endif
#ifdef RWABL_NO_GSTORE
      go = hv == 123.f ? go : OOB;
#endif
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cv), rs_c, (int)co, sc_t + blk * 64, 0);
      if constexpr (SAVE) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 sv = u32x2{pack2(iv, fv), pack2(gv, ov)};
        __builtin_amdgcn_raw_buffer_store_b64(sv, rs_g, (int)go, sg_t + blk * 128, 0);
      }
    };
    // 16-byte chunks k0 .. k1 - 1 of the PREVIOUS step's h rows -> hout (issued between the MFMAs of the step's first block: the
    // tile is rewritten from the second block on)
    auto hout_chunks = [&](int k0, int k1) {
#ifndef RWABL_NO_HOUT
#pragma unroll
      for (int k = k0; k < k1; ++k) {
        if (k < HK) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const uint4 v = *reinterpret_cast<const uint4*>(hs + hlds[k]);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_h, (int)(hoff[k] | hmask), sh_prev, 0);
        }
      }
#endif
    };

    auto body = [&](int bo, auto first) {
#pragma unroll
      for (int bi = 0; bi < RW_UB; ++bi) {
        const int b = bo + bi;
        // accumulators start from the gate pre-activations x W_ih^T + b
        f32x4_t acc[4];
        float cprev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint2 gv2 = gxq[bi][r];
          acc[0][r] = __uint_as_float(gv2.x << 16);
          acc[1][r] = __uint_as_float(gv2.x & 0xffff0000u);
          acc[2][r] = __uint_as_float(gv2.y << 16);
          acc[3][r] = __uint_as_float(gv2.y & 0xffff0000u);
          cprev[r] = cq[bi][r];
        }
        // fetch block b + 3 (of this step, or of the next one: its c_{t-1} is what this wave stored >= 22 blocks ago)
        {
          const int bn = b + RW_PD;
          const bool same = bn < NBLK;
          prefetch((bi + RW_PD) % RW_UB, same ? bn : bn - NBLK, same ? toff : toff_next, same ? toff_prev : toff, same ? step > 0 : true);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int f = 0; f < SF; ++f) {
            const int fb = q * SF + f;                     // fragment within the block: (ks, gate) = (fb / 4, fb % 4)
            const int F = (bi * 4 + q) * SF + f;           // ... within the unrolled body
            const int ks = fb >> 2, gate = fb & 3;
#ifndef RWABL_NO_MFMA
            acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hfrag[ks]),
                                                                __builtin_bit_cast(bf16x8_t, bq[F % RW_D]), acc[gate], 0, 0, 0);
            const int F2 = (F + RW_D) % (RW_UB * NF);      // read-ahead target: stage F2 / 13 -> slot (F2 / 13) % 5
            bq[F % RW_D] = *reinterpret_cast<const uint4*>(ring + ((F2 / SF) % RW_NSLOT) * SLOTB + (F2 % SF) * 1024 + loff);
#endif
          }
#ifndef RWABL_NO_CELL
          if (decltype(first)::value && bi == 0) hout_chunks(q * 4, q * 4 + 4);     // the step's first block: nothing to update yet
          else cell(b - 1, q, std::false_type{});
#endif
          RW_BARRIER();                                    // end of a stage
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) accp[g] = acc[g];
#pragma unroll
        for (int r = 0; r < 4; ++r) cprevp[r] = cprev[r];
      }
    };
    body(0, std::true_type{});
#pragma unroll 1
    for (int bo = RW_UB; bo < NBLK; bo += RW_UB) body(bo, std::false_type{});
    // the last block's cell update has nothing to hide behind: the next step needs all of h_t
#pragma unroll
    for (int r = 0; r < 4; ++r) cell(NBLK - 1, r, std::true_type{});
    // h_t: next step's A fragments
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) hfrag[ks] = *reinterpret_cast<const uint4*>(hs + lc * HPITCH + ks * 64 + 16 * lr);
  }
  // the last step's rows
  {
    const int t = dir ? 0 : p.seq_len - 1;
    const int sh = t * stride_i * ldh_i * 2;
#pragma unroll
    for (int k = 0; k < HK; ++k) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const uint4 v = *reinterpret_cast<const uint4*>(hs + hlds[k]);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_h, (int)hoff[k], sh, 0);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Second form of the same kernel: the unit blocks are PAIRED.  Measured on the first form (profiles/r04_abl_rw_fwd_v3.log): with
// all arithmetic switched off the launch takes as long as with it - the kernel is bound by its own memory pattern, and a
// microbenchmark of that pattern (scripts/diag/stride_bw.hip) says why: 64-byte pieces per row (the f32 cell state of 16 units) load
// at 3.5 TB/s where 256-byte pieces load at 6.2.  Here block 2p holds the EVEN units 32p, 32p + 2, .. and block 2p + 1 the odd ones
// (urse_lstm_pack_blocks_rw permutes the weight columns), so that lane lc owns units 32p + 2 lc and 32p + 2 lc + 1 of a pair:
// its pre-activations are one 16-byte load (256 contiguous bytes per row and wave-instruction), its two cell states one 8-byte
// load (128 bytes per row), and the saved gates / c_t / h_t of a pair leave in one 16- / 8- / 4-byte store each - half the vector
// memory instructions of the first form.  Units 384 .. 391 are a last single block.  Ring slots are addressed with a scalar
// offset (5 slots, 8 stages per pair); everything else - loader, barriers, read-ahead, hout in the step's first block - is as above.
constexpr int RW2_NPAIR = 12;     // pairs of 16-unit blocks (units 0 .. 383); block 24 = units 384 .. 391

template <int H, int HP, bool SAVE>
__global__ void __launch_bounds__(RW_WAVES * 64, 2) lstm_fwd_rw2_kernel(RwArgs p) {
  constexpr int NBLK = (H + 15) / 16, NSLAB = HP / 32, NF = 4 * NSLAB;
  constexpr int SF = NSLAB, SPS = NBLK * 4, SLOTB = SF * 1024, HPITCH = lds_frag_pitch(HP * 2);
  static_assert(NBLK == 2 * RW2_NPAIR + 1 && SF == 13 && H % 8 == 0 && (4 * NF) % RW_D == 0 && NF % RW_D == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;
  char* hsb = smem + RW_NSLOT * SLOTB;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x & 1, wi = blockIdx.x >> 1;
  const int ntl = p.tiles_base + (wi < p.tiles_rem ? 1 : 0);
  const int tile0 = wi * p.tiles_base + min(wi, p.tiles_rem);
  const long total_stages = (long)p.seq_len * SPS;

  if (w == RW_WAVES - 1) {       // loader: identical to the first form's
    const char* wsrc = reinterpret_cast<const char*>(p.whhb) + (long)dir * NBLK * NF * 1024 + lane * 16;
    const unsigned ring0 = (unsigned)(size_t)ring;
    int sm = 0, slot = 0;
    auto issue = [&]() {
      const char* src = wsrc + (long)sm * SLOTB;
      const unsigned dst = ring0 + slot * SLOTB;
#pragma unroll
      for (int f = 0; f < SF; ++f) rw_glds16(src + f * 1024, dst + f * 1024);
      sm = (sm + 1 == SPS) ? 0 : sm + 1;
      slot = (slot + 1 == RW_NSLOT) ? 0 : slot + 1;
    };
    issue(); issue(); issue(); issue();
    asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
    issue();
    __builtin_amdgcn_s_barrier();
    for (long k = 0; k < total_stages; ++k) {
      if (k >= 1) issue();
      asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  if (w >= ntl) return;

  char* hs = hsb + w * 16 * HPITCH;
  const int seq0 = (tile0 + w) * 16;
  for (int i = lane; i < 16 * HPITCH / 16; i += 64) reinterpret_cast<uint4*>(hs)[i] = make_uint4(0, 0, 0, 0);
  constexpr unsigned OOB = 0xFFFFF000u;
  const int ldg_i = (int)p.ldg, ldc_i = 2 * H, ldh_i = (int)p.ldh, gcol_i = dir * 4 * H, hcol_i = dir * H, stride_i = (int)p.stride;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gx, 0, (int)p.gx_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  // per-lane byte offsets of row 4 lr + r at t = 0: pair form (units 2 lc, 2 lc + 1 of pair 0) and tail form (unit 384 + lc)
  // (a row past n_seq gets offsets outside the buffers: its loads return zeros, its stores are dropped)
  unsigned goff[4], coff[4], hsoff[4];
  const bool tvalid = 2 * RW2_NPAIR * 16 + lc < H;
  const unsigned gtd = (unsigned)((2 * RW2_NPAIR * 16 * 4 - lc * 4) * 2), ctd = (unsigned)((2 * RW2_NPAIR * 16 - lc) * 4);   // tail - pair offset
  int* rowtab = reinterpret_cast<int*>(hsb + RW_MAXT * 16 * HPITCH) + w * 16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int seq = seq0 + lr * 4 + r;
    const bool rvalid = seq < p.n_seq;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const unsigned row = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    goff[r] = rvalid ? (row * (unsigned)ldg_i + (unsigned)(gcol_i + lc * 8)) * 2u : OOB;
    coff[r] = rvalid ? (row * (unsigned)ldc_i + (unsigned)(hcol_i + lc * 2)) * 4u : OOB;
    hsoff[r] = (unsigned)((lr * 4 + r) * HPITCH + lc * 4);
  }
  constexpr int CPR = H * 2 / 16, HK = (16 * CPR + 63) / 64;
  if (lane < 16) {                                                      // row of (sequence, t = 0), -1 past n_seq: the hout chunks look it up
    const int seq = seq0 + lane;
    rowtab[lane] = seq < p.n_seq ? (int)((seq / p.inner) * p.outer + (seq % p.inner)) : -1;
  }
  // chunk k of this lane: (row idx / 49, 16-byte piece idx % 49) of the wave's h tile, idx = lane + 64 k
  auto hout_chunk = [&](int k, int soff, unsigned mask) {
    const int idx = lane + 64 * k, row = idx / CPR, cc = idx - row * CPR;
    const int rowc = idx < 16 * CPR ? row : 0;
    const int grow = rowtab[rowc];
    const uint4 v = *reinterpret_cast<const uint4*>(hs + rowc * HPITCH + cc * 16);
    const unsigned off = (idx < 16 * CPR && grow >= 0) ? ((unsigned)grow * (unsigned)ldh_i + (unsigned)(hcol_i + cc * 8)) * 2u : OOB;
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4_{v.x, v.y, v.z, v.w}, rs_h, (int)(off | mask), soff, 0);
  };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto pack2 = [](float a, float b) -> unsigned {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  };

  uint4 hfrag[NSLAB];
#pragma unroll
  for (int ks = 0; ks < NSLAB; ++ks) hfrag[ks] = make_uint4(0, 0, 0, 0);
  u32x4 gq[2][4];            // pre-activations of a pair (two slots): .xy = even unit, .zw = odd unit
  u32x2 cq[2][4];            // c_{t-1} of a pair
  u32x2 gt[4];               // ... of the tail block
  unsigned ct[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    gq[0][r] = gq[1][r] = u32x4{0u, 0u, 0u, 0u};
    cq[0][r] = cq[1][r] = u32x2{0u, 0u};
    gt[r] = u32x2{0u, 0u};
    ct[r] = 0u;
  }
  auto prefetch_pair = [&](int slot, int pr, int toff_, int toffc_, bool has_c) {
    const int sg = (toff_ * ldg_i + pr * 128) * 2, sc = (toffc_ * ldc_i + pr * 32) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gq[slot][r] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)goff[r], sg, 0);
      cq[slot][r] = __builtin_amdgcn_raw_buffer_load_b64(rs_c, (int)(has_c ? coff[r] : OOB), sc, 0);
    }
  };
  auto prefetch_tail = [&](int toff_, int toffc_, bool has_c) {
    const int sg = toff_ * ldg_i * 2, sc = toffc_ * ldc_i * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gt[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)(goff[r] + gtd), sg, 0);       // (OOB + gtd stays outside the buffer)
      ct[r] = __builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(has_c ? coff[r] + ctd : OOB), sc, 0);
    }
  };
  {
    const int t0 = (dir ? p.seq_len - 1 : 0) * stride_i;
    prefetch_pair(0, 0, t0, t0, false);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                         // A_0
  const unsigned loff = lane * 16;
  int so_cur = 0, so_next = SLOTB;                                      // byte offsets of the ring slots of the current / next stage
  uint4 bq[RW_D];
#pragma unroll
  for (int i = 0; i < RW_D; ++i) bq[i] = *reinterpret_cast<const uint4*>(ring + i * 1024 + loff);

  f32x4_t accp[4];                                                      // gates of the previous block, waiting for their cell update
  float cprevp[4];
  float c_e[4], h_e[4];                                                 // results of a pair's even block, held until the odd one is done
  u32x2 g_e[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) accp[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) { cprevp[r] = 0.f; c_e[r] = 0.f; h_e[r] = 0.f; g_e[r] = u32x2{0u, 0u}; }

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const bool more = step + 1 < p.seq_len;
    const int toff_next = more ? (dir ? t - 1 : t + 1) * stride_i : toff;
    const int toff_prev = step > 0 ? (dir ? t + 1 : t - 1) * stride_i : toff;
    const int sg_t = toff * ldg_i * 2, sc_t = toff * ldc_i * 4;
    const int sh_prev = toff_prev * ldh_i * 2;
    const unsigned hmask = step > 0 ? 0u : OOB;

    // the cell of row 4 lr + r from accp / cprevp: (c_t, h_t, packed gates)
    auto cell_math = [&](int r, float& cv, float& hv, u32x2& gs) {
      const float iv = sigmoidf_(accp[0][r]), fv = sigmoidf_(accp[1][r]), gv = tanhf_(accp[2][r]), ov = sigmoidf_(accp[3][r]);
      cv = __builtin_fmaf(fv, cprevp[r], __fmul_rn(iv, gv));
      hv = ov * tanhf_(cv);
      gs = u32x2{pack2(iv, fv), pack2(gv, ov)};
    };
    auto cell_even = [&](int r) { cell_math(r, c_e[r], h_e[r], g_e[r]); };
    auto cell_odd_store = [&](int pr, int r) {                          // second unit of pair `pr`, then the pair's stores
      float cv, hv;
      u32x2 gs;
      cell_math(r, cv, hv, gs);
      *reinterpret_cast<unsigned*>(hs + hsoff[r] + pr * 64) = pack2(h_e[r], hv);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(c_e[r]), __float_as_uint(cv)}, rs_c, (int)coff[r], sc_t + pr * 128, 0);
      if constexpr (SAVE)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{g_e[r][0], g_e[r][1], gs[0], gs[1]}, rs_g, (int)goff[r], sg_t + pr * 256, 0);
    };
    auto cell_tail = [&](int r) {
      float cv, hv;
      u32x2 gs;
      cell_math(r, cv, hv, gs);
      *reinterpret_cast<bf16_t*>(hs + (lr * 4 + r) * HPITCH + (2 * RW2_NPAIR * 16 + lc) * 2) = f32_to_bf16(hv);   // (units past H: k padding)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cv), rs_c, (int)(tvalid ? coff[r] + ctd : OOB), sc_t, 0);
      if constexpr (SAVE) __builtin_amdgcn_raw_buffer_store_b64(gs, rs_g, (int)(tvalid ? goff[r] + gtd : OOB), sg_t, 0);
    };
    auto hout_chunks = [&](int k0, int k1) {
#pragma unroll
      for (int k = k0; k < k1; ++k)
        if (k < HK) hout_chunk(k, sh_prev, hmask);
    };
    // the 52 MFMAs of one block against the ring, four stages with a barrier each; `between(q)` is issued among the MFMAs of quarter q
    auto block_mfma = [&](f32x4_t (&acc)[4], auto between) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int f = 0; f < SF; ++f) {
          const int fb = q * SF + f, ks = fb >> 2, gate = fb & 3;
          acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, hfrag[ks]),
                                                              __builtin_bit_cast(bf16x8_t, bq[fb % RW_D]), acc[gate], 0, 0, 0);
          const int f2 = f + RW_D;                       // read-ahead: this stage's slot, or the head of the next one
          bq[fb % RW_D] = (f2 < SF) ? *reinterpret_cast<const uint4*>(ring + so_cur + f2 * 1024 + loff)
                                    : *reinterpret_cast<const uint4*>(ring + so_next + (f2 - SF) * 1024 + loff);
        }
        between(q);
        __builtin_amdgcn_s_barrier();
        so_cur = so_next;
        so_next = (so_next + SLOTB == RW_NSLOT * SLOTB) ? 0 : so_next + SLOTB;
      }
    };
    auto init_acc = [&](f32x4_t (&acc)[4], unsigned lo, unsigned hi, int r) {
      acc[0][r] = __uint_as_float(lo << 16);
      acc[1][r] = __uint_as_float(lo & 0xffff0000u);
      acc[2][r] = __uint_as_float(hi << 16);
      acc[3][r] = __uint_as_float(hi & 0xffff0000u);
    };
    auto retire = [&](const f32x4_t (&acc)[4], const float (&cprev)[4]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) accp[g] = acc[g];
#pragma unroll
      for (int r = 0; r < 4; ++r) cprevp[r] = cprev[r];
    };

    // two pairs per body: the prefetch slots become constants
    auto body = [&](int p0, auto first) {
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int pr = p0 + pi;
        f32x4_t acc[4];
        float cprev[4];
        // even block
#pragma unroll
        for (int r = 0; r < 4; ++r) { init_acc(acc, gq[pi][r][0], gq[pi][r][1], r); cprev[r] = __uint_as_float(cq[pi][r][0]); }
        block_mfma(acc, [&](int q) {
          if (decltype(first)::value && pi == 0) hout_chunks(q * 4, q * 4 + 4);     // the step's first block: h of the previous step goes out
          else cell_odd_store(pr - 1, q);
        });
        retire(acc, cprev);
        {   // fetch what comes after the odd block (one block = ~3 us ahead): the next pair, or the tail block after the last pair
          if (pr + 1 < RW2_NPAIR) prefetch_pair(pi ^ 1, pr + 1, toff, toff_prev, step > 0);
          else prefetch_tail(toff, toff_prev, step > 0);
        }
        // odd block
#pragma unroll
        for (int r = 0; r < 4; ++r) { init_acc(acc, gq[pi][r][2], gq[pi][r][3], r); cprev[r] = __uint_as_float(cq[pi][r][1]); }
        block_mfma(acc, [&](int q) { cell_even(q); });
        retire(acc, cprev);
      }
    };
    body(0, std::true_type{});
#pragma unroll 1
    for (int p0 = 2; p0 < RW2_NPAIR; p0 += 2) body(p0, std::false_type{});
    // tail block (units 384 ..): the last pair's second unit is updated and the pair stored meanwhile
    {
      prefetch_pair(0, 0, toff_next, toff, true);                        // next step's first pair (c_{t-1} = what this step stored)
      f32x4_t acc[4];
      float cprev[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { init_acc(acc, gt[r][0], gt[r][1], r); cprev[r] = __uint_as_float(ct[r]); }
      block_mfma(acc, [&](int q) { cell_odd_store(RW2_NPAIR - 1, q); });
      retire(acc, cprev);
#pragma unroll
      for (int r = 0; r < 4; ++r) cell_tail(r);
    }
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) hfrag[ks] = *reinterpret_cast<const uint4*>(hs + lc * HPITCH + ks * 64 + 16 * lr);
  }
  {
    const int t = dir ? 0 : p.seq_len - 1;
    const int sh = t * stride_i * ldh_i * 2;
#pragma unroll
    for (int k = 0; k < HK; ++k) hout_chunk(k, sh, 0u);
  }
}

// block-ordered recurrent weights with the row-wave kernel's unit order: block b < 24, column lc = unit 32 (b / 2) + 2 lc + (b & 1);
// block 24, column lc = unit 384 + lc.  Otherwise the layout of urse_lstm_pack_blocks: (dir, blk, ks, gate) = 64 lanes x 16 B,
// lane (lr, lc): k = ks * 32 + 8 lr + j.
__global__ void __launch_bounds__(256) lstm_pack_blocks_rw_kernel(const float* __restrict__ whh, bf16_t* __restrict__ out, int H, int Hp) {
  const int nblk = (H + 15) >> 4, nslab = Hp / 32, G4 = 4 * H, npair = nblk / 2;
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
    const int u = blk < 2 * npair ? 32 * (blk >> 1) + 2 * lc + (blk & 1) : blk * 16 + lc;
    const int k = ks * 32 + 8 * lr + jj;
    out[idx] = f32_to_bf16((u < H && k < H) ? whh[((long)d * G4 + g * H + u) * H + k] : 0.f);
  }
}

static bool rw_shape(int H, int Hp) { return H == 392 && Hp == 416; }

}  // namespace urse

using namespace urse;

extern "C" int urse_lstm_rw_supported(int H, int Hp) { return rw_shape(H, Hp) ? 1 : 0; }

extern "C" int urse_lstm_pack_blocks_rw(const float* whh, void* out, int H, int Hp, void* stream) {
  URSE_CHECK_ARG(whh && out && rw_shape(H, Hp), "urse_lstm_pack_blocks_rw: bad argument (H=%d Hp=%d)", H, Hp);
  hipLaunchKernelGGL(lstm_pack_blocks_rw_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, whh, (bf16_t*)out, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_blocks_rw");
  return URSE_OK;
}

extern "C" int urse_lstm_rw_fwd(void* gx, int64_t ldg, const void* whhb, void* hout, int64_t ldh, float* c, int H, int Hp,
                                int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save,
                                int target_workgroups, int paired, void* stream) {
  URSE_CHECK_ARG(gx && whhb && hout && c, "urse_lstm_rw_fwd: null pointer (c is required: it carries c_{t-1})");
  URSE_CHECK_ARG(rw_shape(H, Hp), "urse_lstm_rw_fwd: unsupported H=%d Hp=%d", H, Hp);
  URSE_CHECK_ARG(n_seq > 0 && seq_len > 0 && inner > 0 && target_workgroups >= 0, "urse_lstm_rw_fwd: bad sequence geometry");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldh < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_rw_fwd: row indices must fit 32 bits");
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 8 == 0 && ldh >= 2L * H && ldh % 8 == 0 && ((uintptr_t)gx % 16) == 0 && ((uintptr_t)hout % 16) == 0 &&
                     ((uintptr_t)c % 8) == 0,
                 "urse_lstm_rw_fwd: bad leading dimension / alignment (hout rows must be 16-byte aligned)");
  RwArgs p;
  p.gx = gx; p.ldg = ldg; p.whhb = whhb; p.hout = hout; p.ldh = ldh; p.c = c; p.save = save;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  {
    // rows the sequence map can touch -> buffer sizes for the range-checked accesses (32-bit byte offsets)
    const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
    const long gb = rows * ldg * 2, cb = rows * 2L * H * 4;
    URSE_CHECK_ARG(gb < 0xFFFFF000L && cb < 0xFFFFF000L, "urse_lstm_rw_fwd: matrices of %ld rows exceed 32-bit byte offsets", rows);
    p.gx_bytes = (unsigned)gb; p.c_bytes = (unsigned)cb; p.h_bytes = (unsigned)(rows * ldh * 2);
  }
  // tiles of 16 sequences per direction, dealt over G workgroups of at most 7: as many workgroups as half the chip (or the
  // caller's target) holds, so that one round covers them with 6-7 tiles each; more tiles than that simply queue
  const int ntile = (n_seq + 15) / 16;
  int half = (target_workgroups > 0 ? target_workgroups : device_cu_count()) / 2;
  if (half < 1) half = 1;
  int G = (ntile + RW_MAXT - 1) / RW_MAXT;
  if (G < half) G = half < ntile ? half : ntile;
  p.tiles_base = ntile / G;
  p.tiles_rem = ntile % G;
  constexpr int HPITCH = lds_frag_pitch(416 * 2);
  const size_t lds = (size_t)RW_NSLOT * 13 * 1024 + (size_t)RW_MAXT * 16 * HPITCH + (paired ? RW_MAXT * 16 * sizeof(int) : 0);
#ifndef URSE_EXPERIMENTS
  // (the paired form - two sequences' gates per lane pair, round 4 - measured no faster than this kernel and spills 11 - 13 registers: variant builds only)
  URSE_CHECK_ARG(!paired, "urse_lstm_rw_fwd: the paired row-wave form is compiled into variant builds only (-DURSE_EXPERIMENTS)");
#else
  if (paired) {
    static bool once2 = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_rw2_kernel<392, 416, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_rw2_kernel<392, 416, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
    (void)once2;
    note_launch(URSE_KV_LSTM_FWD_RW);
    if (save) hipLaunchKernelGGL((lstm_fwd_rw2_kernel<392, 416, true>), dim3(2 * G), dim3(RW_WAVES * 64), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((lstm_fwd_rw2_kernel<392, 416, false>), dim3(2 * G), dim3(RW_WAVES * 64), lds, (hipStream_t)stream, p);
    URSE_CHECK_LAUNCH("urse_lstm_rw_fwd");
    return URSE_OK;
  }
#endif
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_rw_kernel<392, 416, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_rw_kernel<392, 416, false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  note_launch(URSE_KV_LSTM_FWD_RW);
  if (save) hipLaunchKernelGGL((lstm_fwd_rw_kernel<392, 416, true>), dim3(2 * G), dim3(RW_WAVES * 64), lds, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((lstm_fwd_rw_kernel<392, 416, false>), dim3(2 * G), dim3(RW_WAVES * 64), lds, (hipStream_t)stream, p);
  URSE_CHECK_LAUNCH("urse_lstm_rw_fwd");
  return URSE_OK;
}
