// "N-split" LSTM backward-through-time with THREE members and 48 sequences per group (bf16): the time path of BSRNN at C2 (1,088 sequences x 401
// steps per direction; espnet2 BSRNN's rnn_time, reference twin baseline_code/models/bsrnn_flowse.py:296-299).
//
// lstm_nsplit.hip (two workgroups share 32 sequences and split the output columns of dh_rec = dgates x W_hh) is bound by the bytes a CU moves per
// step through its vector memory path: 640 KB of its half of W_hh^T + ~190 KB of inputs, own-half stores and the partner's half = 830 KB at ~65 GB/s
// = 12.8 us, measured 12.5 - 13.7.  Nothing is left to overlap; only fewer bytes per CU and step shorten it.  This kernel re-cuts the
// same idea so that the bytes fall while the CU count stays: THREE workgroups share 48 sequences, member m owns a third of the unit tiles (9 / 8 / 8
// of 25: one tile per wave, nine waves = three per SIMD at most = a 168-register budget, which is what three row tiles of per-row state need) and
// streams only ITS columns of W_hh^T: 0.43 MB per step; with 48 rows of inputs, its third stored and two thirds copied: ~660 KB per CU and step.
// 23 groups x 3 members x 2 directions = 138 workgroups (136 before).  Each fragment of the weight stream feeds three MFMAs instead of two.
//   * per step: gate gradients of the owned units -> LDS tile (own columns) -> barrier -> the own third to the `gates` output as 16-byte row pieces
//     (plain stores when the three members read the same XCC id, write-through otherwise) -> the own K range of the product, behind whose first
//     batch of fragment loads every wave waits for its stores (counted vmcnt) and the last one raises the member's step flag -> wait for BOTH
//     partners' flags, copy their thirds from the gates output into the tile (sc1 loads) -> barrier -> the other two K ranges;
//   * a member waits for two partners, never for a grid: finite work beside the launch can delay a group, not starve it; bounded
//     spins, error flag.  Members of a group are blockIdx.x eight apart (one XCD under round-robin dealing: speed only).
// MEASURED (profiles/r05_exp_nsplit3_v1.log, r05_abl_nsplit3_v1.log; C2 shape, alone): parity as the two-member kernel (gate gradients vs the
// streaming BPTT 3.8e-3 of the scale, error flag 0) - and 5.62 ms per launch against 5.10: SLOWER.  The weight stream alone (everything else switched
// off) takes 8.3 us per step for 427 KB = 51 GB/s, not the 65 GB/s the two-member form reaches with 13 waves; the copy of TWO thirds (104 KB instead of
// 50) costs 3.3 us, the stores 2.3, the input rows of 48 sequences 1.9 - and only a third of the product is left to hide the hand-off behind.  Bytes
// per CU and step fall by 14 %, not by a third, and overlap is lost: the split's fixed costs grow with the member count (what rounds 2 - 4 saw for
// every cluster / split BPTT).  Kept opt-in (URSE_NSPLIT_MEMBERS=3) with its parity test; the two-member kernel ships.
// Same math, layouts and outputs as lstm_bwd_kernel / lstm_bwd_nsplit_kernel; member 0 adds the k-slabs in ascending order (bit-identical with the
// streaming kernel), members 1 and 2 add their own range first.
#include <stdlib.h>

#include "../urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int N3W = 9;             // waves per workgroup: one unit tile each
constexpr int N3THR = N3W * 64;    // 576
constexpr int N3ROWS = 48;         // sequences per group
constexpr int N3RT = N3ROWS / 16;  // row tiles
#ifndef N3_KB
#define N3_KB 11                   // weight fragments in flight per wave (13: four spilled registers, 6.37 instead of 5.62 ms)
#endif
#ifndef N3_PUB
#define N3_PUB 1                   // the batch of the own-range product in front of which the wave publishes
#endif

struct Nsplit3Args {
  const void* dh; long ldd;
  void* gates; long ldg;
  const float* c;
  const void* whhT;                // fragment-ordered [2][nut][nslab][64][16 B] (urse_lstm_pack)
  unsigned* flags;                 // [2 dirs][ngroups][3 members] step flags, then as many XCC-id words; zeroed per launch
  unsigned* err;
  long inner, outer, stride;
  int n_seq, seq_len, ngroups;
  unsigned g_bytes, c_bytes, d_bytes;
};

template <int H>
__global__ void __launch_bounds__(N3THR) lstm_bwd_nsplit3_kernel(Nsplit3Args p) {
  constexpr int NUT = (H + 15) / 16, G4 = 4 * H, NSLAB = G4 * 2 / 64;   // 25 tiles, 49 k-slabs
  constexpr int T0 = (NUT + 2) / 3, T1 = T0 + (NUT - T0 + 1) / 2;       // member 0: tiles [0, 9), 1: [9, 17), 2: [17, 25)
  static_assert(T0 <= N3W && T1 - T0 <= N3W && NUT - T1 <= N3W, "one unit tile per wave");
  constexpr int PITCH = lds_frag_pitch(G4 * 2);
  constexpr int TPR = N3THR / N3ROWS;                                   // 12 threads per row move the row's 16-byte pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                    // [48][PITCH] gate gradients of the step, all units (MFMA A operand)
  unsigned* lsync = reinterpret_cast<unsigned*>(smem + N3ROWS * PITCH); // [0] waves whose stores are complete (monotonic), [1] dead flag, [2] same XCD
  int* rowtab = reinterpret_cast<int*>(lsync + 4);                      // [48] row of (sequence, t = 0), -1 beyond n_seq
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x = (P / 8) * 24 + m * 8 + P % 8: the members of group-and-direction P are eight apart
  const int lin = blockIdx.x, blk = lin / 24, rem = lin - blk * 24, m = rem >> 3, P = blk * 8 + (rem & 7);
  const int dir = P & 1, grp = P >> 1;
  if (grp >= p.ngroups) return;
  const int ut_lo = m == 0 ? 0 : (m == 1 ? T0 : T1), ut_hi = m == 0 ? T0 : (m == 1 ? T1 : NUT);      // owned unit tiles
  const int ut = ut_lo + w;
  const bool active = ut < ut_hi;
  // byte ranges of a row of the tile / of the direction's 4H-column segment of `gates` (16 units x 4 gates x 2 B = 128 B per unit tile)
  const int ob0 = ut_lo * 128, ob1 = (ut_hi * 128 < G4 * 2) ? ut_hi * 128 : G4 * 2;
  const int ks_own0 = 2 * ut_lo, ks_own1 = (2 * ut_hi < NSLAB) ? 2 * ut_hi : NSLAB;        // k-slabs of the owned units' gate columns
  const int u = ut * 16 + lc;
  const bool uvalid = active && u < H;
  const int uc = uvalid ? u : H - 1;
  if (tid < 3) lsync[tid] = 0u;

  // byte offsets at t = 0 of the lane's rows in the C layout (rows past n_seq clamped to the last sequence: loaded, never stored): 32-bit offsets into
  // buffer resources + a scalar per step, instead of 64-bit pointer arithmetic per row and step (registers: the budget is 168)
  unsigned rowq[N3RT][4];                                               // (one register per row; the three matrices' offsets are one 24-bit multiply-add each per step)
  const int s0 = grp * N3ROWS;
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * NUT * NSLAB + (long)(active ? ut : ut_lo) * NSLAB) * 1024 + lane * 16;
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)p.d_bytes, 0x00020000);
#pragma unroll
  for (int rt = 0; rt < N3RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      if (seq >= p.n_seq) seq = p.n_seq - 1;
      const unsigned row = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
      rowq[rt][r] = row;
    }
  const unsigned gbase = (unsigned)(gcol_i + uc * 4) * 2u, cbase = (unsigned)(hcol_i + uc) * 4u, dbase = (unsigned)(hcol_i + uc) * 2u;
  const unsigned ldg2 = (unsigned)ldg_i * 2u, ldc4 = (unsigned)ldc_i * 4u, ldd2 = (unsigned)ldd_i * 2u;      // (< 2^24, as the rows: checked by the host)
  unsigned* fl = p.flags + (dir * p.ngroups + grp) * 3;
  unsigned* my_flag = fl + m;
  const int pa = m == 0 ? 1 : 0, pb = m == 2 ? 1 : 2;                   // the two partners, ascending
  if (tid < N3ROWS) {
    const int seq = s0 + tid;
    rowtab[tid] = seq < p.n_seq ? (int)((seq / p.inner) * p.outer + (seq % p.inner)) : -1;
  }
  for (int i = tid; i < N3ROWS * PITCH / 16; i += N3THR) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);

  float dcs[N3RT][4], dhr[N3RT][4], ccur[N3RT][4];
  {
    const int toff0 = (dir ? 0 : p.seq_len - 1) * stride_i;
#pragma unroll
    for (int rt = 0; rt < N3RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dcs[rt][r] = 0.f;
        dhr[rt][r] = 0.f;
        ccur[rt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(__umul24(rowq[rt][r], ldc4) + cbase), toff0 * ldc_i * 4, 0));
      }
  }
  // Same XCD?  Each member publishes the XCC id it READS from the hardware and reads its partners': a group on one XCD shares that XCD's L2, so its
  // gate gradients can be handed over with plain stores (acknowledged by the L2, kept there for the partners' L1-bypassing loads) instead of
  // write-through ones.  Never inferred from blockIdx.
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;      // HW_REG_XCC_ID
    unsigned* xw = p.flags + 2 * 3 * p.ngroups + (dir * p.ngroups + grp) * 3;
    __hip_atomic_store(xw + m, xcc | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned same = 1u;
    for (int q = 0; q < 3; ++q) {
      if (q == m) continue;
      unsigned v = 0u, spins = 0;
      while ((v = __hip_atomic_load(xw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }
      }
      if (v != (xcc | 0x100u)) same = 0u;
    }
    lsync[2] = same;
  }
  __syncthreads();
  const bool local = lsync[2] != 0u;
  bool dead = false;
  // the thread's row of the tile for the 16-byte pieces it moves (own third out, partners' thirds in): row tid / 12, pieces tid % 12 + 12 j
  const int mv_row = tid / TPR, mv_c0 = tid - mv_row * TPR;
  const int mv_grow = rowtab[mv_row];
  const unsigned mv_lds = (unsigned)(mv_row * PITCH + mv_c0 * 16);
  const unsigned mv_glb = mv_grow >= 0 ? (unsigned)(((long)mv_grow * ldg_i + gcol_i) * 2 + mv_c0 * 16) : 0xFFFFF000u;      // (a row past n_seq: loads return zeros, stores are dropped)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int MAXP = (T0 * 128 / 16 + TPR - 1) / TPR;                 // 6 rounds of 12 pieces cover the widest third (72 pieces)

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? step : (p.seq_len - 1 - step);
    const int toff = t * stride_i;
    const bool first_ = dir ? (t == p.seq_len - 1) : (t == 0);          // first step of the forward recurrence: c_{-1} = 0
    const bool last = step + 1 == p.seq_len;
    // (the row registers are made opaque per step: visible as loop invariants, the 36 offsets derived from them are hoisted out of the time loop and
    //  spilled)
#pragma unroll
    for (int rt = 0; rt < N3RT; ++rt) asm volatile("" : "+v"(rowq[rt][0]), "+v"(rowq[rt][1]), "+v"(rowq[rt][2]), "+v"(rowq[rt][3]));
    // ---- 1. gate gradients of the owned units -> LDS tile (own columns)
    if (active) {
#pragma unroll
      for (int rt = 0; rt < N3RT; ++rt) {
        uint2 gpre[4];
        float cpre[4];
        bf16_t dhpre[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef N3ABL_NO_LOAD      // timing diagnostics (wrong results): N3ABL_NO_LOAD, N3ABL_NO_STORE, N3ABL_NO_MM, N3ABL_NO_POLL, N3ABL_NO_COPY
          gpre[r] = make_uint2(rowq[rt][r], 0x3f003f00u); cpre[r] = (float)(toff & 3); dhpre[r] = (bf16_t)(0x3c00 + (toff & 7));
#else
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 gv2 = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)(__umul24(rowq[rt][r], ldg2) + gbase), toff * ldg_i * 2, 0);
          gpre[r] = make_uint2(gv2[0], gv2[1]);
          cpre[r] = first_ ? 0.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(__umul24(rowq[rt][r], ldc4) + cbase), (toff + prev_i) * ldc_i * 4, 0));
          dhpre[r] = (bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_d, (int)(__umul24(rowq[rt][r], ldd2) + dbase), toff * ldd_i * 2, 0);
#endif
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float iv = __uint_as_float(gpre[r].x << 16), fv = __uint_as_float(gpre[r].x & 0xffff0000u);
          const float gv = __uint_as_float(gpre[r].y << 16), ov = __uint_as_float(gpre[r].y & 0xffff0000u);
          const float dht = bf16_to_f32(dhpre[r]) + dhr[rt][r];
          const float tc = tanhf_(ccur[rt][r]);
          const float dct = dcs[rt][r] + dht * ov * (1.f - tc * tc);
          const float d0 = dct * gv * iv * (1.f - iv), d1 = dct * cpre[r] * fv * (1.f - fv);
          const float d2 = dct * iv * (1.f - gv * gv), d3 = dht * tc * ov * (1.f - ov);
          dcs[rt][r] = dct * fv;
          ccur[rt][r] = cpre[r];                                         // c_{t-1} is the next processed step's c_t
          uint2 pk = make_uint2(0u, 0u);
          if (uvalid) {
            pk.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
            pk.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
          }
          if (uvalid) *reinterpret_cast<uint2*>(tile + (rt * 16 + lr * 4 + r) * PITCH + u * 8) = pk;     // (columns past 4H stay zero)
        }
      }
    }
    // (raw barriers: __syncthreads() would also drain the stores below - their latency belongs behind the weight stream)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // the own third of the tile is complete
    // the own third of the tile -> the gates output, 16 bytes per lane along the rows
    const unsigned step_off = (unsigned)((long)toff * ldg_i * 2);
#ifndef N3ABL_NO_STORE
#pragma unroll
    for (int jj = 0; jj < MAXP; ++jj) {
      const int cb = ob0 + (mv_c0 + TPR * jj) * 16;
      if (cb < ob1) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + mv_lds + ob0 + TPR * 16 * jj);
        if (local) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)(mv_glb + (unsigned)(ob0 + TPR * 16 * jj)), (int)step_off, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)(mv_glb + (unsigned)(ob0 + TPR * 16 * jj)), (int)step_off, 16);
      }
    }
#endif
    if (last) break;                                                     // (the last step's gradients are stored; nothing waits for them)
    f32x4_t acc[N3RT];
#pragma unroll
    for (int rt = 0; rt < N3RT; ++rt) acc[rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* ar = tile + lc * PITCH + 16 * lr;
    // k-slabs [k0, k1) of this wave's unit tile against the tile in LDS, N3_KB fragments in flight; `publish`: after the batch N3_PUB of fragment
    // loads is issued, wait for this wave's gate-gradient stores (older in the in-order vmcnt queue) and count the wave in
    auto product = [&](int k0, int k1, bool publish) {
#pragma unroll 1
      for (int kb = k0; kb < k1; kb += N3_KB) {
        uint4 b[N3_KB];
#pragma unroll
        for (int i = 0; i < N3_KB; ++i) {
          const int ks = (kb + i < k1) ? kb + i : k1 - 1;
          b[i] = *reinterpret_cast<const uint4*>(whhT + (long)ks * 1024);
        }
        if (publish && (kb == k0 + N3_PUB * N3_KB || (kb == k0 && k0 + N3_PUB * N3_KB >= k1))) {
          // this batch's N3_KB fragment loads may stay in flight; everything older - the earlier batches and, before them, this wave's stores - is
          // then complete
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N3_KB) : "memory");
          if (lane == 0) {
            const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
            if (n == (unsigned)N3W * (unsigned)(step + 1))               // every wave of this member has waited for its stores
              __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#ifndef N3ABL_NO_MM
#pragma unroll
        for (int i = 0; i < N3_KB; ++i) {
          if (kb + i < k1) {
#pragma unroll
            for (int rt = 0; rt < N3RT; ++rt) {
              const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + (kb + i) * 64);
              acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b[i]), acc[rt], 0, 0, 0);
            }
          }
        }
#else
        acc[0][0] += __uint_as_float(b[0].x);
#endif
      }
    };
    // ---- 2. the own K range (in LDS already); the hand-off to the partners travels meanwhile
    if (active) {
      product(ks_own0, ks_own1, true);
    } else {                                                             // a wave without a unit tile stored its share of the rows too
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
        if (n == (unsigned)N3W * (unsigned)(step + 1)) __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- 3. the partners' thirds: wait for both flags, copy their columns of the 48 rows from the gates output into the tile
    if (w == 0 && lane == 0) {
      unsigned spins = 0;
#ifndef N3ABL_NO_POLL
      while (!dead && (__hip_atomic_load(fl + pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1) ||
                       __hip_atomic_load(fl + pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1))) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { dead = true; atomicExch(p.err, 1u); lsync[1] = 1u; }
      }
#endif
    }
    __syncthreads();
#ifndef N3ABL_NO_COPY
    {
      // bytes [0, ob0) and [ob1, 4H * 2) of the row: pieces mv_c0 + 12 j of each range (the two ranges are adjacent for members 0 and 2)
      constexpr int MAXC = (T1 * 128 / 16 + TPR - 1) / TPR;              // rounds over the widest "everything but the own third" (member 2: bytes [0, 128 T1))
      u32x4 v[MAXC];
#pragma unroll
      for (int jj = 0; jj < MAXC; ++jj) {
        int cb = (mv_c0 + TPR * jj) * 16;                                // byte in the row, skipping the own range
        if (cb >= ob0) cb += ob1 - ob0;
        v[jj] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)(cb < G4 * 2 ? mv_glb - (unsigned)(mv_c0 * 16) + (unsigned)cb : 0xFFFFF000u), (int)step_off, 16);      // sc1: L1-bypassing
      }
#pragma unroll
      for (int jj = 0; jj < MAXC; ++jj) {
        int cb = (mv_c0 + TPR * jj) * 16;
        if (cb >= ob0) cb += ob1 - ob0;
        if (cb < G4 * 2) *reinterpret_cast<uint4*>(tile + mv_row * PITCH + cb) = make_uint4(v[jj][0], v[jj][1], v[jj][2], v[jj][3]);
      }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- 4. the other K ranges, ascending
    if (active) {
      if (ks_own0 > 0) product(0, ks_own0, false);
      if (ks_own1 < NSLAB) product(ks_own1, NSLAB, false);
#pragma unroll
      for (int rt = 0; rt < N3RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[rt][r] = acc[rt][r];
    }
    // no barrier here: the next step's cell phase rewrites the OWN columns, whose last readers (this step's own-range product) sit behind two
    // barriers; the next copy rewrites the PARTNERS' columns behind the next step's two barriers
  }
}

}  // namespace urse

using namespace urse;

// -> plan {groups per direction, workgroups, flag words}; < 0 (URSE_ERR_UNSUPPORTED) if the shape has no kernel or the groups would not be
// co-resident beside the reserved CUs
extern "C" int urse_lstm_nsplit3_plan(int H, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && n_seq > 0 && reserved_cus >= 0, "urse_lstm_nsplit3_plan: bad argument");
  if (H != 392) {
    set_error("urse_lstm_nsplit3_plan: unsupported H=%d", H);
    return URSE_ERR_UNSUPPORTED;
  }
  const int ngroups = (n_seq + N3ROWS - 1) / N3ROWS;
  const int wgs = ((2 * ngroups + 7) / 8) * 24;
  if (wgs > device_cu_count() - reserved_cus) {
    set_error("urse_lstm_nsplit3_plan: %d workgroups do not fit beside %d reserved CUs", wgs, reserved_cus);
    return URSE_ERR_UNSUPPORTED;
  }
  plan[0] = ngroups; plan[1] = wgs; plan[2] = 2L * (2L * ngroups * 3);      // step flags + XCC ids
  return URSE_OK;
}

extern "C" int urse_lstm_nsplit3_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT, void* flags,
                                     void* err_flag, int H, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                                     int reserved_cus, void* stream) {
  URSE_CHECK_ARG(dh && gates && c && whhT && flags && err_flag, "urse_lstm_nsplit3_bwd: null pointer");
  int64_t plan[3];
  int rc = urse_lstm_nsplit3_plan(H, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(seq_len > 0 && inner > 0 && ldg >= 8L * H && ldd >= 2L * H && ldg % 8 == 0 && ((uintptr_t)gates % 16) == 0,
                 "urse_lstm_nsplit3_bwd: bad leading dimension / alignment");
  const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
  URSE_CHECK_ARG(rows * ldg * 2 < 0xFFFFF000L && rows * 2L * H * 4 < 0xFFFFF000L && rows * ldd * 2 < 0xFFFFF000L && ldd < (1L << 31),
                 "urse_lstm_nsplit3_bwd: the gates / c / dh matrices exceed 32-bit byte offsets");
  URSE_CHECK_ARG(rows < (1L << 24) && ldg * 2 < (1L << 24) && ldd * 2 < (1L << 24), "urse_lstm_nsplit3_bwd: rows / leading dimensions exceed 24 bits");
  Nsplit3Args p;
  p.dh = dh; p.ldd = ldd; p.gates = gates; p.ldg = ldg; p.c = c; p.whhT = whhT; p.flags = (unsigned*)flags; p.err = (unsigned*)err_flag;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len; p.ngroups = (int)plan[0];
  p.g_bytes = (unsigned)(rows * ldg * 2); p.c_bytes = (unsigned)(rows * 2L * H * 4); p.d_bytes = (unsigned)(rows * ldd * 2);
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(flags, 0, sizeof(unsigned) * plan[2], st);
  const size_t lds = (size_t)N3ROWS * lds_frag_pitch(4 * 392 * 2) + 16 + N3ROWS * sizeof(int);
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_nsplit3_kernel<392>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                160 * 1024), true);
  (void)once;
  note_launch(URSE_KV_LSTM_BWD_NSPLIT3);
  hipLaunchKernelGGL((lstm_bwd_nsplit3_kernel<392>), dim3((unsigned)plan[1]), dim3(N3THR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_nsplit3_bwd");
  return URSE_OK;
}
