// Row-wave LSTM forward with the INPUT PROJECTION FUSED (bf16): the band path of BSRNN at C2 (12,832 sequences x 34 steps per
// direction; espnet2 BSRNN's rnn_freq, reference twin baseline_code/models/bsrnn_flowse.py:303-306, nn.LSTM = x W_ih^T + b_ih + h W_hh^T + b_hh).
//
// lstm_rw.hip (a wave owns 16 sequences, the seven compute waves of a workgroup share one pass over W_hh streamed L2 -> LDS by a
// loader wave) turned out to be bound by its own memory traffic: with every MFMA and every transcendental switched off the launch
// takes as long as with them (profiles/r04_abl_rw_fwd_v3.log), and a third of that traffic is the gate pre-activations gx = x W_ih^T + b,
// which the gate-projection GEMM wrote (2.74 GB per launch) a millisecond earlier only to be read back here.  The matrix pipe is
// idle, so the projection is done HERE: W_ih (196 -> 224 input channels = 7 more k-slabs per block) rides on the same LDS ring
// behind the 13 slabs of W_hh, the wave's 16 rows of the normalised input x_n (448 bytes each) are seven more resident A fragments,
// the accumulators start from the bias.  No gx matrix is written or read any more; the gates buffer is written once, with the
// activations the backward needs.  Per launch: the 0.94 ms gate GEMM disappears and the recurrence moves 5.4 GB instead of 8.9.
// Numerics: the pre-activation is no longer rounded to bf16 between the two products (one f32 accumulator through all 20 slabs) -
// closer to the f32 reference than the two-kernel form, not bit-identical with it.
// Geometry: 80 fragments per 16-unit block (20 slabs x 4 gates) = 8 stages of 10 KB, ring of 6 slots, four stages in flight;
// everything else - loader protocol, barriers, read-ahead of 4 fragments, cell update of block b - 1 between the MFMAs of block b,
// c_{t-1} round trip through the f32 stream, hout in the step's first block, buffer addressing with out-of-range offsets for lanes
// that must not store - is lstm_rw.hip's.
#include <stdlib.h>

#include <type_traits>

#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int RX_MAXT = 7, RX_WAVES = 8;
constexpr int RX_NSLOT = 6;       // ring slots
constexpr int RX_SF = 10;         // fragments per stage
constexpr int RX_UB = 5;          // blocks per unrolled body
constexpr int RX_D = 4;           // fragments read ahead
constexpr int RX_PD = 3;          // blocks c_{t-1} / the bias are fetched ahead
#ifndef RX_LOADER
#define RX_LOADER 0                // 0: the loader wave feeds the ring by LDS-DMA, 1: through registers (global load + LDS store) - measured equal, see below
#endif
#ifndef RX_NIF
#define RX_NIF 4                   // stages (10 fragments each) the loader keeps in flight in registers
#endif

struct RxArgs {
  const void* xn; long ldx;       // [M, ldx] bf16 normalised input, K padding zero
  const void* wx;                 // [2][NBLK][20 slabs][4 gates][64 lanes][16 B]   (urse_lstm_pack_blocks_x)
  const float* bias;              // [2][4H] f32, (unit, gate) interleaved: b_ih + b_hh
  void* gates; long ldg;          // out (save): gate activations, (unit, gate) interleaved
  void* hout; long ldh;
  void* hout2;                    // f16 operands: h once more in bf16 for the weight-gradient GEMMs (null: not wanted)
  float* c;
  long inner, outer, stride;
  int n_seq, seq_len;
  unsigned x_bytes, g_bytes, c_bytes, h_bytes;
  int tiles_base, tiles_rem;
};

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void rx_glds16(const char* gsrc, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

// TI: operand format of both products (bf16_t | f16_t: x_n, W_ih, W_hh, the carried h and hout); the saved gate activations are bf16 in both
// (they feed the BPTT); H2: also write hout2 (f16 only)
#ifdef RXSTAMP
__device__ unsigned long long g_rxstamps[64 * 4];
#endif
template <int H, int HP, int NP, bool SAVE, typename TI = bf16_t, bool H2 = false>
__global__ void __launch_bounds__(RX_WAVES * 64, 2) lstm_fwd_rwx_kernel(RxArgs p) {
  static_assert(!H2 || __is_same(TI, f16_t), "the bf16 copy of h exists in the f16 mode only");
  constexpr int NBLK = (H + 15) / 16, NSH = HP / 32, NSX = NP / 32, NS = NSH + NSX, NF = 4 * NS;   // 25 blocks, 13 + 7 slabs, 80 fragments per block
  constexpr int SF = RX_SF, SPB = NF / SF, SPS = NBLK * SPB, SLOTB = SF * 1024;              // 8 stages per block, 200 per step
  constexpr int HPITCH = lds_frag_pitch(HP * 2);
  static_assert(NF % SF == 0 && NF % RX_D == 0 && SF == 10 && NBLK % RX_UB == 0 && SPB == 8 && H % 8 == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                                    // [NSLOT][SF][1 KiB]
  char* hsb = smem + RX_NSLOT * SLOTB;                                  // [MAXT][16][HPITCH]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x & 1, wi = blockIdx.x >> 1;
  const int ntl = p.tiles_base + (wi < p.tiles_rem ? 1 : 0);
  const int tile0 = wi * p.tiles_base + min(wi, p.tiles_rem);
  const long total_stages = (long)p.seq_len * SPS;

  if (w == RX_WAVES - 1) {
    // loader: stage s -> slot s % 6.  Invariant at barrier A_k: stages <= k + 1 have landed, the compute waves are done with stages <= k - 1;
    // after A_k stage k + 5 goes into the slot stage k - 1 has left; four stages (40 KB) in flight.
    const char* wsrc = reinterpret_cast<const char*>(p.wx) + (long)dir * NBLK * NF * 1024 + lane * 16;
#if RX_LOADER == 1
    // Round 5 experiment (off): the ring fed through REGISTERS - global_load_dwordx4 into a rotating set of 4 x 10 fragments, ds_write_b128 into the slot
    // just in time.  The question it answered: one wave issues 2,000 LDS-DMA instructions per step, and the in-kernel stamps of lstm_clusterx.hip priced
    // such an instruction at 100 - 300 cycles of its wave - 2,000 x 100 cycles IS the 87 us step; is this launch bound by the loader's issue rate rather
    // than by the CU's memory path (round 4's reading)?  No: a load + an LDS store cost the wave ~25 cycles per fragment and the launch takes the same
    // 3.07 ms (profiles/r05_abl_rwx_loader_v2.log).  Bytes through the CU, whoever issues them.
    // Invariant at barrier A_k as before: stages <= k + 1 are in LDS, the compute waves are done with stages <= k - 1.
    static_assert(RX_NIF * SF <= 60, "counted waits");
    static_assert(RX_NIF >= 4 && RX_NIF <= 6 && SF == 10, "the loader's rotation below is written out for four to six stages of ten fragments in flight");
    // (named registers, written out: as an array indexed through unrolled loops inside lambdas the buffer stayed in scratch memory)
#define RX_F10(M, s) M(s, 0) M(s, 1) M(s, 2) M(s, 3) M(s, 4) M(s, 5) M(s, 6) M(s, 7) M(s, 8) M(s, 9)
#define RX_DECL(s, f) uint4 fb_##s##_##f;
    RX_F10(RX_DECL, 0) RX_F10(RX_DECL, 1) RX_F10(RX_DECL, 2) RX_F10(RX_DECL, 3)
#if RX_NIF > 4
    RX_F10(RX_DECL, 4)
#endif
#if RX_NIF > 5
    RX_F10(RX_DECL, 5)
#endif
    int sm = 0;                                                          // stage (within the step) the next loads fetch
    const char* src_;
    char* dst_;
#define RX_LD1(s, f) fb_##s##_##f = *reinterpret_cast<const uint4*>(src_ + (f) * 1024);
#define RX_ST1(s, f) *reinterpret_cast<uint4*>(dst_ + (f) * 1024) = fb_##s##_##f;
#define RX_GLOAD(s) do { src_ = wsrc + (long)sm * SLOTB; RX_F10(RX_LD1, s) sm = (sm + 1 == SPS) ? 0 : sm + 1; } while (0)
#define RX_LWRITE(s, slot_) do { dst_ = ring + (slot_) * SLOTB + lane * 16; RX_F10(RX_ST1, s) } while (0)
    // prologue: stages 0, 1 into LDS; stages 2 .. 1 + NIF in flight
    RX_GLOAD(0); RX_GLOAD(1);
    RX_LWRITE(0, 0); RX_LWRITE(1, 1);
    RX_GLOAD(0); RX_GLOAD(1); RX_GLOAD(2); RX_GLOAD(3);                 // fb_i <- stage 2 + i
#if RX_NIF > 4
    RX_GLOAD(4);
#endif
#if RX_NIF > 5
    RX_GLOAD(5);
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // A_0
    int slot = 2;                                                        // slot of stage k + 2
    // k: stage k + 2 sits in fb_s -> its slot (the compiler waits for exactly these ten loads: the younger ones stay in flight), then stage k + 2 + NIF is
    // requested into the same registers (past the end: wraps into the weights again, harmless)
#define RX_TURN(s) do { RX_LWRITE(s, slot); slot = (slot + 1 == RX_NSLOT) ? 0 : slot + 1; RX_GLOAD(s); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                        __builtin_amdgcn_s_barrier(); } while (0)
    long k0 = 0;
    for (; k0 + RX_NIF <= total_stages; k0 += RX_NIF) {
      RX_TURN(0); RX_TURN(1); RX_TURN(2); RX_TURN(3);
#if RX_NIF > 4
      RX_TURN(4);
#endif
#if RX_NIF > 5
      RX_TURN(5);
#endif
    }
    {                                                                    // the remaining total_stages % NIF turns
      const int left = (int)(total_stages - k0);
      if (left > 0) RX_TURN(0);
      if (left > 1) RX_TURN(1);
      if (left > 2) RX_TURN(2);
      if (left > 3) RX_TURN(3);
#if RX_NIF > 5
      if (left > 4) RX_TURN(4);
#endif
    }
#undef RX_TURN
#undef RX_GLOAD
#undef RX_LWRITE
#undef RX_LD1
#undef RX_ST1
#undef RX_DECL
#undef RX_F10
    return;
#else
    const unsigned ring0 = (unsigned)(size_t)ring;
    int sm = 0, slot = 0;
    auto issue = [&]() {
      const char* src = wsrc + (long)sm * SLOTB;
      const unsigned dst = ring0 + slot * SLOTB;
#ifndef RXABL_NO_DMA
#pragma unroll
      for (int f = 0; f < SF; ++f) rx_glds16(src + f * 1024, dst + f * 1024);
#endif
      sm = (sm + 1 == SPS) ? 0 : sm + 1;
      slot = (slot + 1 == RX_NSLOT) ? 0 : slot + 1;
    };
    issue(); issue(); issue(); issue(); issue();
    asm volatile("s_waitcnt vmcnt(30)" ::: "memory");                   // stages 0, 1 landed
    issue();
    __builtin_amdgcn_s_barrier();                                       // A_0
#ifdef RXSTAMP     // timing diagnostics (scripts/stamps.py): the LOADER wave of workgroup RXSTAMP sees both sides of the ring - per time step (200 stages) the
                   // shader-clock cycles it spent issuing the ten LDS-DMAs of a stage, waiting for stage k + 2 to LAND (vmcnt), and waiting at the stage's
                   // BARRIER for the seven compute waves; [step][4] = {issue, landing wait, barrier wait, stages}.  (The compute waves cannot be stamped
                   // without breaking their read-ahead: s_memtime is a scalar memory instruction and its wait drains the LDS reads in flight.)
    unsigned long long a_issue = 0, a_land = 0, a_bar = 0;
    const bool stamp_on = blockIdx.x == RXSTAMP;
#endif
    for (long k = 0; k < total_stages; ++k) {
#ifdef RXSTAMP
      const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
      if (k >= 1) issue();                                              // stage k + 5
#ifdef RXSTAMP
      const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
#ifndef RXABL_NO_DMA_WAIT
      asm volatile("s_waitcnt vmcnt(30)" ::: "memory");                 // stage k + 2 landed
#endif
#ifdef RXSTAMP
      const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();                                     // A_{k+1}
#ifdef RXSTAMP
      const unsigned long long t3_ = __builtin_amdgcn_s_memtime();
      a_issue += t1_ - t0_; a_land += t2_ - t1_; a_bar += t3_ - t2_;
      if ((k + 1) % SPS == 0) {
        const long st_ = k / SPS;
        if (stamp_on && lane == 0 && st_ < 64) {
          g_rxstamps[st_ * 4 + 0] = a_issue; g_rxstamps[st_ * 4 + 1] = a_land; g_rxstamps[st_ * 4 + 2] = a_bar; g_rxstamps[st_ * 4 + 3] = SPS;
        }
        a_issue = a_land = a_bar = 0;
      }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
#endif
  }
  if (w >= ntl) return;

  // ---------------- compute wave: 16 sequences ----------------
  char* hs = hsb + w * 16 * HPITCH;
  int* rowtab = reinterpret_cast<int*>(hsb + RX_MAXT * 16 * HPITCH) + w * 16;
  const int seq0 = (tile0 + w) * 16;
  for (int i = lane; i < 16 * HPITCH / 16; i += 64) reinterpret_cast<uint4*>(hs)[i] = make_uint4(0, 0, 0, 0);
  constexpr unsigned OOB = 0xFFFFF000u;
  const int ldg_i = (int)p.ldg, ldc_i = 2 * H, ldh_i = (int)p.ldh, ldx_i = (int)p.ldx;
  const int gcol_i = dir * 4 * H, hcol_i = dir * H, stride_i = (int)p.stride;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h2 = __builtin_amdgcn_make_buffer_rsrc(H2 ? p.hout2 : p.hout, 0, (int)p.h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.xn), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, 2 * 4 * H * 4, 0x00020000);
  // per-lane byte offsets at t = 0, block 0: rows 4 lr + r (C layout: cell update, stores), row lc (A layout: x fragments)
  unsigned goff[4], coff[4], hsoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int seq = seq0 + lr * 4 + r;
    const bool rvalid = seq < p.n_seq;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const unsigned row = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    goff[r] = rvalid ? (row * (unsigned)ldg_i + (unsigned)(gcol_i + lc * 4)) * 2u : OOB;      // (a row past n_seq: loads return zeros, stores are dropped)
    coff[r] = rvalid ? (row * (unsigned)ldc_i + (unsigned)(hcol_i + lc)) * 4u : OOB;
    hsoff[r] = (unsigned)((lr * 4 + r) * HPITCH + lc * 2);
  }
  unsigned xoff;
  {
    int seq = seq0 + lc;
    const bool rvalid = seq < p.n_seq;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const unsigned row = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    xoff = rvalid ? (row * (unsigned)ldx_i) * 2u + (unsigned)(16 * lr) : OOB;
  }
  const unsigned boff = (unsigned)((gcol_i + lc * 4) * 4);                // bias of (unit lc, 4 gates) of block 0
  if (lane < 16) {
    const int seq = seq0 + lane;
    rowtab[lane] = seq < p.n_seq ? (int)((seq / p.inner) * p.outer + (seq % p.inner)) : -1;
  }
  constexpr int CPR = H * 2 / 16, HK = (16 * CPR + 63) / 64;            // 49 chunks of 16 B per h row, 13 chunks per lane
  auto hout_chunk = [&](int k, int soff, unsigned mask) {
    const int idx = lane + 64 * k, row = idx / CPR, cc = idx - row * CPR;
    const int rowc = idx < 16 * CPR ? row : 0;
    const int grow = rowtab[rowc];
    const uint4 v = *reinterpret_cast<const uint4*>(hs + rowc * HPITCH + cc * 16);
    const unsigned off = (idx < 16 * CPR && grow >= 0) ? ((unsigned)grow * (unsigned)ldh_i + (unsigned)(hcol_i + cc * 8)) * 2u : OOB;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_h, (int)(off | mask), soff, 0);
    if constexpr (H2) {
      float a0, a1, a2, a3, a4, a5, a6, a7;
      unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7)},
                                             rs_h2, (int)(off | mask), soff, 0);
    }
  };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  auto pack2b = [](float a, float b) -> unsigned {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
  };

  uint4 hfrag[NSH], xfrag[NSX];                                         // A fragments: h_{t-1} and x_t of the wave's 16 rows
#pragma unroll
  for (int ks = 0; ks < NSH; ++ks) hfrag[ks] = make_uint4(0, 0, 0, 0);
  auto load_x = [&](int toff_) {
    const int sx = toff_ * ldx_i * 2;
#pragma unroll
    for (int ks = 0; ks < NSX; ++ks) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xoff, sx + ks * 64, 0);
      xfrag[ks] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  float cq[RX_UB][4];                                                   // c_{t-1}, slot = block % 5
  f32x4_t bq4[RX_UB];                                                   // bias of the block's four gates
#pragma unroll
  for (int a = 0; a < RX_UB; ++a) {
    bq4[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) cq[a][r] = 0.f;
  }
  auto prefetch = [&](int slot, int blk, int toffc_, bool has_c) {
    const int sc = (toffc_ * ldc_i + blk * 16) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#ifdef RXABL_NO_CLOAD
      cq[slot][r] = __uint_as_float((unsigned)sc & 0x3fffffffu);
#else
      cq[slot][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(has_c ? coff[r] : OOB), sc, 0));
#endif
    const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)boff, blk * 256, 0);
    bq4[slot] = f32x4_t{__uint_as_float(bv[0]), __uint_as_float(bv[1]), __uint_as_float(bv[2]), __uint_as_float(bv[3])};
  };
  {
    const int t0 = (dir ? p.seq_len - 1 : 0) * stride_i;
    load_x(t0);
#pragma unroll
    for (int b = 0; b < RX_PD; ++b) prefetch(b, b, t0, false);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                         // A_0: stages 0 and 1 are in the ring
  const unsigned loff = lane * 16;
  int so_cur = 0, so_next = SLOTB;
  uint4 bq[RX_D];
#pragma unroll
  for (int i = 0; i < RX_D; ++i) bq[i] = *reinterpret_cast<const uint4*>(ring + i * 1024 + loff);

  f32x4_t accp[4];
  float cprevp[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) accp[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) cprevp[r] = 0.f;

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const bool more = step + 1 < p.seq_len;
    const int toff_next = more ? (dir ? t - 1 : t + 1) * stride_i : toff;
    const int toff_prev = step > 0 ? (dir ? t + 1 : t - 1) * stride_i : toff;
    const int sg_t = toff * ldg_i * 2, sc_t = toff * ldc_i * 4;
    const int sh_prev = toff_prev * ldh_i * 2;
    const unsigned hmask = step > 0 ? 0u : OOB;

    auto cell = [&](int blk, int r, auto tail) {
#ifdef RXABL_CHEAP_CELL
      const float iv = accp[0][r], fv = accp[1][r], gv = accp[2][r], ov = accp[3][r];
      const float cv = __builtin_fmaf(fv, cprevp[r], __fmul_rn(iv, gv));
      const float hv = ov * cv;
#else
      const float iv = sigmoidf_(accp[0][r]), fv = sigmoidf_(accp[1][r]), gv = tanhf_(accp[2][r]), ov = sigmoidf_(accp[3][r]);
      const float cv = __builtin_fmaf(fv, cprevp[r], __fmul_rn(iv, gv));
      const float hv = ov * tanhf_(cv);
#endif
      *reinterpret_cast<TI*>(hs + hsoff[r] + blk * 32) = from_f32<TI>(hv);      // (units past H: the tile's k padding, finite values against zero weights)
      unsigned co = coff[r], go = goff[r];
#ifdef RXABL_NO_STORE
      co = hv == 123.f ? co : OOB; go = hv == 123.f ? go : OOB;
#endif
#ifdef RXABL_CHEAP_CELL
      (void)0;
#endif
      if constexpr (decltype(tail)::value) {
        const bool uvalid = blk * 16 + lc < H;
        co = uvalid ? co : OOB;
        go = uvalid ? go : OOB;
      }
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cv), rs_c, (int)co, sc_t + blk * 64, 0);
      if constexpr (SAVE)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack2b(iv, fv), pack2b(gv, ov)}, rs_g, (int)go, sg_t + blk * 128, 0);
    };

    auto body = [&](int bo, auto first) {
#pragma unroll
      for (int bi = 0; bi < RX_UB; ++bi) {
        const int b = bo + bi;
        f32x4_t acc[4];
        float cprev[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4_t{bq4[bi][g], bq4[bi][g], bq4[bi][g], bq4[bi][g]};      // b_ih + b_hh of (unit lc, gate g)
#pragma unroll
        for (int r = 0; r < 4; ++r) cprev[r] = cq[bi][r];
        {
          const int bn = b + RX_PD;
          const bool same = bn < NBLK;
          prefetch((bi + RX_PD) % RX_UB, same ? bn : bn - NBLK, same ? toff_prev : toff, same ? step > 0 : true);
        }
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
#pragma unroll
          for (int f = 0; f < SF; ++f) {
            const int fb = q * SF + f, ks = fb >> 2, gate = fb & 3;       // slabs 0 .. 12: h W_hh^T, 13 .. 19: x W_ih^T
            const uint4 a = ks < NSH ? hfrag[ks < NSH ? ks : 0] : xfrag[ks >= NSH ? ks - NSH : 0];
#ifndef RXABL_NO_MFMA
            acc[gate] = mfma16<TI>(a, bq[fb % RX_D], acc[gate]);
#else
            acc[gate][0] += __uint_as_float(a.x ^ bq[fb % RX_D].x);
#endif
            const int f2 = f + RX_D;
            bq[fb % RX_D] = (f2 < SF) ? *reinterpret_cast<const uint4*>(ring + so_cur + f2 * 1024 + loff)
                                      : *reinterpret_cast<const uint4*>(ring + so_next + (f2 - SF) * 1024 + loff);
          }
          if (decltype(first)::value && bi == 0) {                       // the step's first block: h of the previous step goes out (13 chunks)
#pragma unroll
            for (int k = q * 2; k < q * 2 + 2; ++k)
              if (k < HK) hout_chunk(k, sh_prev, hmask);
          } else if ((q & 1) == 0) {
#ifndef RXABL_NO_CELL
            cell(b - 1, q >> 1, std::false_type{});
#endif
          }
          __builtin_amdgcn_s_barrier();
          so_cur = so_next;
          so_next = (so_next + SLOTB == RX_NSLOT * SLOTB) ? 0 : so_next + SLOTB;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) accp[g] = acc[g];
#pragma unroll
        for (int r = 0; r < 4; ++r) cprevp[r] = cprev[r];
      }
    };
    body(0, std::true_type{});
#pragma unroll 1
    for (int bo = RX_UB; bo < NBLK; bo += RX_UB) body(bo, std::false_type{});
    load_x(toff_next);                                                    // x of the next step: in flight during the tail below
#pragma unroll
    for (int r = 0; r < 4; ++r) cell(NBLK - 1, r, std::true_type{});     // the last block's update has nothing to hide behind
#pragma unroll
    for (int ks = 0; ks < NSH; ++ks) hfrag[ks] = *reinterpret_cast<const uint4*>(hs + lc * HPITCH + ks * 64 + 16 * lr);
  }
  {
    const int t = dir ? 0 : p.seq_len - 1;
    const int sh = t * stride_i * ldh_i * 2;
#pragma unroll
    for (int k = 0; k < HK; ++k) hout_chunk(k, sh, 0u);
  }
}

// Block-ordered weights of the fused kernel: (dir, blk, slab, gate) = 64 lanes x 16 B; lane (lr, lc): unit blk * 16 + lc;
// slab ks < Hp / 32: W_hh[gate * H + unit][ks * 32 + 8 lr + j]; the following Np / 32 slabs: W_ih[gate * H + unit][(ks - Hp / 32) * 32 + 8 lr + j];
// zeros past H / N.
template <typename TI>
__device__ __forceinline__ void lstm_pack_blocks_x_dev(const float* __restrict__ wih, const float* __restrict__ whh,
                                                       TI* __restrict__ out, int N, int Np, int H, int Hp) {
  const int nblk = (H + 15) >> 4, nsh = Hp / 32, ns = nsh + Np / 32, G4 = 4 * H;
  const long total = (long)2 * nblk * ns * 4 * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int jj = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int g = (int)(r % 4); r /= 4;
    const int ks = (int)(r % ns); r /= ns;
    const int blk = (int)(r % nblk);
    const int d = (int)(r / nblk);
    const int lc = lane & 15, lr = lane >> 4;
    const int u = blk * 16 + lc;
    float v = 0.f;
    if (u < H) {
      if (ks < nsh) {
        const int k = ks * 32 + 8 * lr + jj;
        if (k < H) v = whh[((long)d * G4 + g * H + u) * H + k];
      } else {
        const int k = (ks - nsh) * 32 + 8 * lr + jj;
        if (k < N) v = wih[((long)d * G4 + g * H + u) * N + k];
      }
    }
    out[idx] = from_f32<TI>(v);
  }
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_blocks_x_kernel(const float* __restrict__ wih, const float* __restrict__ whh,
                                                                 TI* __restrict__ out, int N, int Np, int H, int Hp) {
  lstm_pack_blocks_x_dev<TI>(wih, whh, out, N, Np, H, Hp);
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_blocks_x_multi_kernel(const PackRow* __restrict__ tab, int N, int Np, int H, int Hp) {
  const PackRow r = tab[blockIdx.y];
  if (r.wx) lstm_pack_blocks_x_dev<TI>(r.wih, r.whh, (TI*)r.wx, N, Np, H, Hp);
}

static bool rx_shape(int N, int Np, int H, int Hp) { return N == 196 && Np == 224 && H == 392 && Hp == 416; }

}  // namespace urse

using namespace urse;

extern "C" int urse_lstm_rwx_supported(int N, int Np, int H, int Hp) { return rx_shape(N, Np, H, Hp) ? 1 : 0; }

extern "C" int urse_lstm_pack_blocks_x(const float* wih, const float* whh, void* out, int N, int Np, int H, int Hp, int dtype, void* stream) {
  URSE_CHECK_ARG(wih && whh && out && rx_shape(N, Np, H, Hp) && (dtype == URSE_BF16 || dtype == URSE_F16),
                 "urse_lstm_pack_blocks_x: bad argument (N=%d Np=%d H=%d Hp=%d dtype=%d)", N, Np, H, Hp, dtype);
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_blocks_x_kernel<f16_t>, dim3(512), dim3(256), 0, (hipStream_t)stream, wih, whh, (f16_t*)out, N, Np, H, Hp);
  else hipLaunchKernelGGL(lstm_pack_blocks_x_kernel<bf16_t>, dim3(512), dim3(256), 0, (hipStream_t)stream, wih, whh, (bf16_t*)out, N, Np, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_blocks_x");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_blocks_x_multi(const void* table, int n_lstm, int N, int Np, int H, int Hp, int dtype, void* stream) {
  URSE_CHECK_ARG(table && n_lstm > 0 && n_lstm < 65536 && rx_shape(N, Np, H, Hp) && (dtype == URSE_BF16 || dtype == URSE_F16),
                 "urse_lstm_pack_blocks_x_multi: bad argument (N=%d Np=%d H=%d Hp=%d dtype=%d)", N, Np, H, Hp, dtype);
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_blocks_x_multi_kernel<f16_t>, dim3(512, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H, Hp);
  else hipLaunchKernelGGL(lstm_pack_blocks_x_multi_kernel<bf16_t>, dim3(512, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, N, Np, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_blocks_x_multi");
  return URSE_OK;
}

extern "C" int urse_lstm_rwx_fwd(const void* xn, int64_t ldx, const void* wx, const float* bias, void* gates, int64_t ldg, void* hout,
                                 int64_t ldh, float* c, int N, int Np, int H, int Hp, int n_seq, int seq_len, int64_t inner,
                                 int64_t outer, int64_t stride, int save, int target_workgroups, int dtype, void* hout_bf16, void* stream) {
  URSE_CHECK_ARG(xn && wx && bias && hout && c && (gates || !save), "urse_lstm_rwx_fwd: null pointer (c is required: it carries c_{t-1})");
  URSE_CHECK_ARG(dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_rwx_fwd: operands are bf16 or f16 (dtype %d)", dtype);
  URSE_CHECK_ARG(!hout_bf16 || (dtype == URSE_F16 && ((uintptr_t)hout_bf16 % 16) == 0), "urse_lstm_rwx_fwd: the bf16 copy of h goes with f16 operands only");
  URSE_CHECK_ARG(rx_shape(N, Np, H, Hp), "urse_lstm_rwx_fwd: unsupported N=%d Np=%d H=%d Hp=%d", N, Np, H, Hp);
  URSE_CHECK_ARG(n_seq > 0 && seq_len > 0 && inner > 0 && target_workgroups >= 0, "urse_lstm_rwx_fwd: bad sequence geometry");
  URSE_CHECK_ARG(ldx >= Np && ldx % 8 == 0 && ((uintptr_t)xn % 16) == 0 && ldh >= 2L * H && ldh % 8 == 0 && ((uintptr_t)hout % 16) == 0 &&
                     (!save || (ldg >= 8L * H && ldg % 4 == 0 && ((uintptr_t)gates % 8) == 0)) && ((uintptr_t)bias % 16) == 0,
                 "urse_lstm_rwx_fwd: bad leading dimension / alignment");
  RxArgs p;
  p.xn = xn; p.ldx = ldx; p.wx = wx; p.bias = bias; p.gates = gates ? gates : hout; p.ldg = save ? ldg : 8L * H; p.hout = hout; p.hout2 = hout_bf16; p.ldh = ldh; p.c = c;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  {
    const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
    const long gb = rows * p.ldg * 2, cb = rows * 2L * H * 4, hb = rows * ldh * 2, xb = rows * ldx * 2;
    URSE_CHECK_ARG(gb < 0xFFFFF000L && cb < 0xFFFFF000L && hb < 0xFFFFF000L && xb < 0xFFFFF000L,
                   "urse_lstm_rwx_fwd: matrices of %ld rows exceed 32-bit byte offsets", rows);
    p.g_bytes = save ? (unsigned)gb : 0u; p.c_bytes = (unsigned)cb; p.h_bytes = (unsigned)hb; p.x_bytes = (unsigned)xb;
  }
  const int ntile = (n_seq + 15) / 16;
  int half = (target_workgroups > 0 ? target_workgroups : device_cu_count()) / 2;
  if (half < 1) half = 1;
  int G = (ntile + RX_MAXT - 1) / RX_MAXT;
  if (G < half) G = half < ntile ? half : ntile;
  p.tiles_base = ntile / G;
  p.tiles_rem = ntile % G;
  constexpr int HPITCH = lds_frag_pitch(416 * 2);
  const size_t lds = (size_t)RX_NSLOT * RX_SF * 1024 + (size_t)RX_MAXT * 16 * HPITCH + RX_MAXT * 16 * sizeof(int);
#define URSE_RX_ATTR(...) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_rwx_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  static bool once = (URSE_RX_ATTR(392, 416, 224, true), URSE_RX_ATTR(392, 416, 224, false), URSE_RX_ATTR(392, 416, 224, true, f16_t, false),
                      URSE_RX_ATTR(392, 416, 224, true, f16_t, true), URSE_RX_ATTR(392, 416, 224, false, f16_t, false), true);
  (void)once;
#undef URSE_RX_ATTR
  note_launch(URSE_KV_LSTM_FWD_RWX);
  const dim3 grid(2 * G), blk(RX_WAVES * 64);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == URSE_F16) {
    if (save && hout_bf16) hipLaunchKernelGGL((lstm_fwd_rwx_kernel<392, 416, 224, true, f16_t, true>), grid, blk, lds, st, p);
    else if (save) hipLaunchKernelGGL((lstm_fwd_rwx_kernel<392, 416, 224, true, f16_t, false>), grid, blk, lds, st, p);
    else hipLaunchKernelGGL((lstm_fwd_rwx_kernel<392, 416, 224, false, f16_t, false>), grid, blk, lds, st, p);
  } else if (save) hipLaunchKernelGGL((lstm_fwd_rwx_kernel<392, 416, 224, true>), grid, blk, lds, st, p);
  else hipLaunchKernelGGL((lstm_fwd_rwx_kernel<392, 416, 224, false>), grid, blk, lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_rwx_fwd");
  return URSE_OK;
}

#ifdef RXSTAMP
extern "C" int urse_diag_rwx_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(urse::g_rxstamps), sizeof(unsigned long long) * 64 * 4);
}
#endif
