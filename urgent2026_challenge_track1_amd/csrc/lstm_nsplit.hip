// "N-split" LSTM backward-through-time (bf16) for few, long sequences: the time path of BSRNN at C2 (1,088 sequences x 401 steps per
// direction; espnet2 BSRNN's rnn_time, reference twin baseline_code/models/bsrnn_flowse.py:296-299).
//
// lstm_bwd_kernel gives 16 sequences to a workgroup, which re-streams all of W_hh^T (1.23 MB) from L2 every step: 10.3 of the step's
// 17.6 us at 90 % of one CU's L2 port - the loop is at its floor, only a different division of labour helps.
// Here TWO workgroups (a pair, one CU each, 136 workgroups in all as before) share 32 sequences and split the OUTPUT columns of the
// recurrent product dh_rec[32, H] = dgates[32, 4H] x W_hh:
//   * member m owns the hidden units of its half (m = 0: unit tiles 0 .. 12, m = 1: tiles 13 .. 24; one tile per wave, both
//     16-row tiles of the pair's sequences) - it keeps dc / dh_rec / c for those units only, forms the gate gradients of those
//     units and streams only ITS columns of W_hh^T: 0.64 MB per step instead of 1.23;
//   * the product needs the gate gradients of ALL units (the reduction runs over 4H).  Each member writes its half to the `gates`
//     output (which the weight-gradient GEMMs read after the launch anyway) with write-through (sc1) stores and raises a flag;
//     the partner copies that half from there into its LDS tile.  The hand-off is issued BEFORE the member multiplies its own
//     half (K range of its own units, already in LDS), so its latency hides behind ~2.5 us of weight stream; only then does the
//     member wait for the partner's flag, copy 50 KB (L1-bypassing loads) and multiply the other half;
//   * protocol (MI355X_MICROARCH.md, "Valid forms", row 1): every payload store sc1; every storing wave waits for its own stores
//     (a counted vmcnt that leaves the weight fragments issued meanwhile in flight) and then adds to a counter in LDS; the wave
//     whose add completes the count stores the flag (sc1); the consumer's wave 0 polls the flag with sc1 loads, a workgroup
//     barrier follows, every load of the payload is an sc1 load.  Placement-independent; the two members of a pair are blockIdx.x
//     eight apart, which under the round-robin dealing of workgroups puts them on ONE XCD (speed only).  Bounded spins, error flag.
// Tried and dropped (profiles/r04_ab_nsplit_band_v1.log): (i) roles taken by ARRIVAL (a ticket counter) instead of blockIdx, which would let the
// grid exceed the chip: the tickets do not follow the round-robin dealing, partners land on different XCDs and the launch goes from 5.24 to
// 5.6 ms (37.2 instead of 33.5 ms per step); (ii) this kernel on the band path (12,832 sequences = 1,604 workgroups): 3.50 against 3.85 ms
// alone, but 28.3 against 25.6 ms per step beside the weight-gradient GEMMs of the second queue.
// Same math, layouts and outputs as lstm_bwd_kernel; member 0 adds the k-slabs in ascending order (bit-identical), member 1 adds
// its own (upper) range first, i.e. the f32 sums of its units see the slabs in another order.
#include <type_traits>

#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int NSW = 13;            // waves per workgroup: one unit tile each
constexpr int NSTHR = NSW * 64;
#ifndef NS_PUB
#define NS_PUB 1                  // the batch of the own-range product in front of which the wave publishes (0 .. 2)
#endif
constexpr int NS_KB = 9;           // weight fragments in flight per wave (13 waves: 128-VGPR cap)
#ifndef NS_PFI
#define NS_PFI 0                  // round 6 EXPERIMENT (compile-time, off: scripts/build_variant.sh pfi lstm_nsplit "-DNS_PFI=1"): the NEXT step's inputs requested before the
#endif                            // second product of this step - the saved gates of the wave's unit tile by LDS-DMA into 4 KB per wave (the 52 KB the tile leaves free),
                                  // c_{t-1} and dh into 16 registers - so that the cell phase starts on data that is there (round 5's stamps: 6,000 + 2,500..4,400 of a
                                  // step's 30,400 cycles are that one exposed round trip + its barrier).  Parity-green, no spill at 6 fragments in flight in the second
                                  // product - and SLOWER: 6.23 against 5.65 ms per launch, 131.8 against 129.2 ms per step, same box both orders
                                  // (profiles/r06_ab_nsplit_pfi_v1.log).  A wave's loads return in order: the twenty prefetch loads (HBM latency) sit in front of the
                                  // second product's weight fragments (L2 hits), which now wait for them - the round trip moved from the cell phase into the product,
                                  // and the resident fragments + three fragments in flight were paid for it.  Only ANOTHER wave's queue could carry the prefetch, and the
                                  // LDS a helper would fill (gates 53 + c 27 + dh 13 KB) is not there (58 KB free).
#ifndef NS_RES
#define NS_RES (NS_PFI ? 0 : 4)   // weight fragments per wave that stay in LDS for the whole launch (the first slabs of the partner's K range): -0.3 ms per train step
#endif                            // in round 5; the prefetch above takes the same LDS and pays more
#ifndef NS_KB2
#define NS_KB2 (NS_PFI ? 6 : 9)   // fragments in flight in the SECOND product, across which the prefetched c / dh registers are live (128-register cap)
#endif

// one LDS-DMA wave-instruction (64 lanes x 16 B, lane-linear at the LDS byte address dst; m0 declared clobbered, as in gemm.hip)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void ns_glds16(const char* gsrc, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(dst) : "memory", "m0");
}
#pragma clang diagnostic pop

struct NsplitArgs {
  const void* dh; long ldd;
  void* gates; long ldg;
  const float* c;
  const void* whhT;                // fragment-ordered [2][nut][nslab][64][16 B] (urse_lstm_pack)
  unsigned* flags;                 // [2 dirs][npairs][2 members] step flags, then as many XCC-id words; zeroed per launch
  unsigned* err;
  long inner, outer, stride;
  int n_seq, seq_len, npairs;
  unsigned g_bytes, c_bytes, d_bytes;
};

// HELP > 0: that many HELPER waves beside the 13 compute waves.  The helpers own the hand-off: after the barrier that completes the own
// half of the tile they store it (16-byte row pieces), wait for those stores, raise the flag, wait for the partner's flag and copy the
// partner's half into the tile - all of it WHILE the compute waves multiply the own K range, which then starts on a vmcnt queue with no
// store in it.  The ablation of the helper-less form (profiles/r04_abl_nsplit_v4.log) priced the own-half stores at 1.2 us and the copy at
// 1.2 us of a 12.7 us step, both in series with the products; 16 waves = 4 per SIMD, the 128-VGPR cap the 13-wave form already had.
#ifdef NSSTAMP      // timing diagnostics: cycle stamps of workgroup NSSTAMP, waves 0 and 6: [step][2][8] (scripts/abl_nsplit.py)
__device__ unsigned long long g_nsstamps[512 * 16];
#define NST_(slot) do { if (stamp_on && step < 512) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_nsstamps[step * 16 + (w ? 8 : 0) + (slot)] = t_; } } while (0)
#else
#define NST_(slot) do { } while (0)
#endif

// TCH = 1: one more wave that does nothing but TOUCH the next step's input rows (one dword per 64-byte sector of the gates / c / dh segments the
// member's cell phase will read), so that those loads find their lines in the XCD's L2: the in-kernel stamps (profiles/r05_abl_nsplit_stamps_v1.log) put
// 6,000 of a step's 30,400 cycles on the cell phase - one exposed HBM round trip per step - and 2,500 - 4,400 more on the barrier behind it.  The loads
// have to come from a wave of their own: in a compute wave's in-order vmcnt queue an HBM-latency load holds back every younger L2 load.
// MEASURED (opt-in, URSE_NSPLIT_TOUCH=1; profiles/r05_exp_nsplit_touch_v1.log): bit-identical and 6.41 against 5.07 ms - the touches are 94 KB more per step
// through the SAME CU's memory path (830 -> 924 KB), which is what the step is made of; a warm L2 does not pay for them.  (Round 3 had priced the same idea
// for the streaming kernel at -4 % with the lines warmed for free.)
template <int H, int HELP, int TCH = 0>
__global__ void __launch_bounds__((NSW + HELP + TCH) * 64) lstm_bwd_nsplit_kernel(NsplitArgs p) {
  constexpr int NTHR_ALL = (NSW + HELP + TCH) * 64;
  constexpr int NUT = (H + 15) / 16, G4 = 4 * H, NSLAB = G4 * 2 / 64, UT0 = (NUT + 1) / 2;     // 25 tiles, 49 slabs, member 0 owns 13 tiles
  constexpr int PITCH = lds_frag_pitch(G4 * 2);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                   // [32][PITCH] gate gradients of the step, all units (MFMA A operand)
  unsigned* lsync = reinterpret_cast<unsigned*>(smem + 32 * PITCH);    // [0] waves whose stores are complete (monotonic), [1] dead flag
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x = (P / 8) * 16 + m * 8 + P % 8: the members of pair-and-direction P are eight apart (same XCD under round-robin dealing)
  const int lin = blockIdx.x, m = (lin >> 3) & 1, P = (lin >> 4) * 8 + (lin & 7);
  const int dir = P & 1, pair = P >> 1;
  if (pair >= p.npairs) return;
#ifdef NSSTAMP
  const bool stamp_on = blockIdx.x == NSSTAMP && (w == 0 || w == 6);
#endif
  const int ut_lo = m ? UT0 : 0, ut_hi = m ? NUT : UT0;                 // owned unit tiles
  const int ut = ut_lo + w;
  const bool active = ut < ut_hi;
  const bool helper = HELP > 0 && w >= NSW;
  const int hidx = (w - NSW) * 64 + lane;                               // helper lane index 0 .. 64 HELP - 1
  const int ks_own0 = 2 * ut_lo, ks_own1 = (2 * ut_hi < NSLAB) ? 2 * ut_hi : NSLAB;        // k-slabs of the owned units' gate columns
  const int u = ut * 16 + lc;
  const bool uvalid = active && u < H;
  const int uc = uvalid ? u : H - 1;
  if (tid < 3) lsync[tid] = 0u;

  int rowbase[2][4];                                                     // row of (sequence, t = 0); negative: beyond n_seq (clamped, never stored)
  const int s0 = pair * 32;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      const bool ok = seq < p.n_seq;
      if (!ok) seq = p.n_seq - 1;
      const int rb = (int)((seq / p.inner) * p.outer + (seq % p.inner));
      rowbase[rt][r] = ok ? rb : -rb - 1;
    }
  auto rowb = [&](int rt, int r) -> int { return rowbase[rt][r] >= 0 ? rowbase[rt][r] : -(rowbase[rt][r] + 1); };
  const char* whhT = reinterpret_cast<const char*>(p.whhT) + ((long)dir * NUT * NSLAB + (long)(active ? ut : ut_lo) * NSLAB) * 1024 + lane * 16;
  const bf16_t* dh = reinterpret_cast<const bf16_t*>(p.dh);
  bf16_t* gates = reinterpret_cast<bf16_t*>(p.gates);
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  unsigned* my_flag = p.flags + ((dir * p.npairs + pair) * 2 + m);
  unsigned* partner_flag = p.flags + ((dir * p.npairs + pair) * 2 + (m ^ 1));
  // the partner's half of a row of the tile: bytes [pb0, pb1) of the direction's 4H-column segment, in 16-byte chunks
  const int pb0 = m ? 0 : UT0 * 128, pb1 = m ? UT0 * 128 : G4 * 2, pcpr = (pb1 - pb0) / 16;
  // the rows of the pair's sequences (for the copy of the partner's half): row of sequence i at t = 0, -1 beyond n_seq
  int* rowtab = reinterpret_cast<int*>(lsync + 4);
  if (tid < 32) {
    const int seq = s0 + tid;
    rowtab[tid] = seq < p.n_seq ? (int)((seq / p.inner) * p.outer + (seq % p.inner)) : -1;
  }
  for (int i = tid; i < 32 * PITCH / 16; i += NTHR_ALL) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
  // NS_PFI: 4 KB per wave behind the tile: [32 rows][16 units x 4 gates] bf16 of the wave's unit tile, the next step's saved gate activations
  char* pfg = smem + 32 * PITCH + 256 + (w < NSW ? w : 0) * 4096;
  // resident weight fragments: the first NS_RES k-slabs of the PARTNER's K range of this wave's unit tile, read from LDS every step
  char* resw = smem + 32 * PITCH + 256 + (w < NSW ? w : 0) * (NS_RES * 1024) + lane * 16;
  if (NS_RES > 0 && active) {
    const int k0o = m ? 0 : ks_own1;
#pragma unroll
    for (int i = 0; i < NS_RES; ++i) *reinterpret_cast<uint4*>(resw + i * 1024) = *reinterpret_cast<const uint4*>(whhT + (long)(k0o + i) * 1024);
  }

  float dcs[2][4], dhr[2][4], ccur[2][4];
  {
    const int toff0 = (dir ? 0 : p.seq_len - 1) * stride_i;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dcs[rt][r] = 0.f;
        dhr[rt][r] = 0.f;
        ccur[rt][r] = p.c[(long)(rowb(rt, r) + toff0) * ldc_i + (hcol_i + uc)];
      }
  }
  // Same XCD?  Each member publishes the XCC id it READS from the hardware and reads its partner's: a pair on one XCD shares that XCD's
  // L2, so its gate gradients can be handed over with plain stores (acknowledged by the L2, kept there for the partner's L1-bypassing
  // loads) instead of write-through ones (acknowledged by memory: ~2 us that every later wait of the storing wave inherits through the
  // in-order vmcnt).  A pair on two XCDs keeps the write-through stores.  Never inferred from blockIdx.
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;      // HW_REG_XCC_ID
    unsigned* xw = p.flags + 2 * 2 * p.npairs + ((dir * p.npairs + pair) * 2);
    __hip_atomic_store(xw + m, xcc | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v = 0u, spins = 0;
    while ((v = __hip_atomic_load(xw + (m ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }
    }
    lsync[2] = (v == (xcc | 0x100u)) ? 1u : 0u;
  }
  __syncthreads();
  const bool local = lsync[2] != 0u;
  bool dead = false;
  if constexpr (HELP > 0) {
    if (helper) {
      // ---- the helper waves' own time loop (same barriers as the compute waves': one after the cell phase, one after the hand-off) ----
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const int ob0 = m ? UT0 * 128 : 0;
      // helper lanes: the 16-byte pieces this lane moves each step, as (tile offset, gates offset at t = 0) without the half's column base;
      // one mapping for both halves (104 pieces per row, the wider half): bit i of h_own / h_par says whether piece i exists in that half
      constexpr int HCPR = UT0 * 8, HCH = HELP ? (32 * HCPR + 64 * HELP - 1) / (64 * HELP) : 1;
      int h_lds[HCH];
      unsigned h_glb[HCH];
      unsigned h_own = 0u, h_par = 0u;
      {
        const int ocpr_ = m ? (G4 * 2 - UT0 * 128) / 16 : HCPR, pcpr_ = m ? HCPR : (G4 * 2 - UT0 * 128) / 16;
    #pragma unroll
        for (int i = 0; i < HCH; ++i) {
          const int idx = hidx + i * 64 * HELP;
          const int row = idx / HCPR, cc = idx - row * HCPR;
          int seq = s0 + row;
          const bool rok = helper && row < 32 && seq < p.n_seq;
          if (!rok) seq = 0;
          const long grow = (seq / p.inner) * p.outer + (seq % p.inner);
          h_lds[i] = (row & 31) * PITCH + cc * 16;
          h_glb[i] = (unsigned)((grow * ldg_i + gcol_i) * 2 + cc * 16);
          if (rok && cc < ocpr_) h_own |= 1u << i;
          if (rok && cc < pcpr_) h_par |= 1u << i;
        }
      }

      for (int step = 0; step < p.seq_len; ++step) {
        const int t = dir ? step : (p.seq_len - 1 - step);
        const unsigned step_off = (unsigned)((long)t * stride_i * ldg_i * 2);
        __builtin_amdgcn_s_barrier();                                    // the own half of the tile is complete
#ifndef NSABL_NO_STORE
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
          if ((h_own >> i) & 1u) {
            const uint4 v = *reinterpret_cast<const uint4*>(tile + h_lds[i] + ob0);
            const unsigned off = h_glb[i] + step_off + (unsigned)ob0;
            if (local) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)off, 0, 0);       // stays in the pair's L2
            else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)off, 0, 16);           // sc1: write-through
          }
        }
#endif
        if (step + 1 == p.seq_len) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
          const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
          if (n == (unsigned)HELP * (unsigned)(step + 1))                // every helper wave has waited for its stores
            __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spins = 0;
#ifdef NSABL_NO_POLL
          while (false && __hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
#else
          while (!dead && __hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
#endif
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { dead = true; atomicExch(p.err, 1u); lsync[1] = 1u; }
          }
        }
        __builtin_amdgcn_wave_barrier();
#ifndef NSABL_NO_COPY
        u32x4 v[HCH];
#pragma unroll
        for (int i = 0; i < HCH; ++i) {
          const unsigned off = ((h_par >> i) & 1u) ? h_glb[i] + step_off + (unsigned)pb0 : 0xFFFFF000u;     // (outside the buffer: returns 0, never written)
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)off, 0, 16);                               // sc1: L1-bypassing
        }
#pragma unroll
        for (int i = 0; i < HCH; ++i)
          if ((h_par >> i) & 1u) *reinterpret_cast<uint4*>(tile + h_lds[i] + pb0) = make_uint4(v[i][0], v[i][1], v[i][2], v[i][3]);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // the partner's half is in the tile
      }
      return;
    }
  }

  if constexpr (TCH > 0) {
    if (w == NSW + HELP) {
      // sectors of a row: the member's own bytes of the gates segment, of c and of dh
      const int ob0t = m ? UT0 * 128 : 0, obnt = (m ? G4 * 2 : UT0 * 128) - ob0t;                     // gates: bytes [ob0t, ob0t + obnt)
      const int ngs = (obnt + 63) / 64, ncs = (obnt / 2 + 63) / 64, nds = (obnt / 4 + 63) / 64, nsec = ngs + ncs + nds;      // (c: 4 B per unit, dh: 2 B: obnt / 8 units)
      const char* gbase = reinterpret_cast<const char*>(p.gates) + (long)gcol_i * 2 + ob0t;
      const char* cbase = reinterpret_cast<const char*>(p.c) + ((long)hcol_i + ob0t / 8) * 4;
      const char* dbase = reinterpret_cast<const char*>(p.dh) + ((long)hcol_i + ob0t / 8) * 2;
      unsigned sink = 0u;      // ONE register that every touch load writes, kept live to the end: the loads land whenever they land, and a register the compiler
                               // believed free (an address, say) must not be what they land in
      for (int step = 0; step < p.seq_len; ++step) {
        __builtin_amdgcn_s_barrier();                                    // (the compute waves' barrier behind the cell phase)
        if (step + 1 == p.seq_len) break;
        if (step + 2 <= p.seq_len) {
          const int tn = dir ? step + 1 : (p.seq_len - 2 - step);        // the NEXT step's time index
          const long toffn = (long)tn * stride_i;
          const bool firstn = dir ? (tn == p.seq_len - 1) : (tn == 0);     // (that step reads no c_{t-1})
          for (int idx = lane; idx < 32 * nsec; idx += 64) {
            const int row = idx / nsec, sec = idx - row * nsec;
            const int grow = rowtab[row];
            if (grow < 0) continue;
            const char* a;
            if (sec < ngs) a = gbase + (grow + toffn) * ldg_i * 2 + sec * 64;
            else if (sec < ngs + ncs) { if (firstn) continue; a = cbase + (grow + toffn + prev_i) * (long)ldc_i * 4 + (sec - ngs) * 64; }
            else a = dbase + (grow + toffn) * ldd_i * 2 + (sec - ngs - ncs) * 64;
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(a) : "memory");      // never waited for: the line is what is wanted
          }
        }
        __builtin_amdgcn_s_barrier();                                    // (behind the poll: the compute waves' __syncthreads; raw here - a fence would wait for the touches)
        __builtin_amdgcn_s_barrier();                                    // (behind the copy)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" :: "v"(sink));
      return;
    }
  }
  // ---- NS_PFI: the inputs of time index tt for this wave's unit tile: gates -> pfg by LDS-DMA (four instructions: lane l of instruction i moves the 16-byte
  // piece l % 8 of row 8 i + l / 8, i.e. lane-linear in the wave's 4 KB), c_{t-1} and dh -> cn / dhn.  Issued in front of the second product's fragment loads:
  // older than every one of them in the in-order vmcnt queue, so complete when the last fragment is.
  float cn[2][4];
  bf16_t dhn[2][4];
  const unsigned pfg_u = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)pfg);
  auto prefetch_inputs = [&](int tt) __attribute__((always_inline)) {
    if (!active) {          // (assigned on every path: a value carried over from the previous step would be live through the whole step - 16 registers)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cn[rt][r] = 0.f; dhn[rt][r] = 0; }
      return;
    }
    const int toffn = tt * stride_i;
    const bool firstn = dir ? (tt == p.seq_len - 1) : (tt == 0);
    // (the tile of the last 8 units: its pieces 4 .. 7 would lie past the direction's 4H columns - they re-read piece 3, nobody reads them back)
    const int pc = (ut * 16 + (lane & 7) * 2 < H) ? (lane & 7) : ((H - ut * 16) / 2 - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int grow = rowtab[i * 8 + (lane >> 3)];
      grow = grow < 0 ? rowtab[0] : grow;                                // (rows past n_seq: any row of the pair, never stored)
      const char* src = reinterpret_cast<const char*>(gates) + ((long)(grow + toffn) * ldg_i + (gcol_i + ut * 64)) * 2 + pc * 16;
      ns_glds16(src, __builtin_amdgcn_readfirstlane(pfg_u + (unsigned)i * 1024u));
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowb(rt, r) + toffn;
        cn[rt][r] = firstn ? 0.f : p.c[(long)(row + prev_i) * ldc_i + (hcol_i + uc)];
        dhn[rt][r] = dh[(long)row * ldd_i + (hcol_i + uc)];
      }
  };
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { cn[rt][r] = 0.f; dhn[rt][r] = 0; }
  if constexpr (NS_PFI && HELP == 0 && TCH == 0) prefetch_inputs(dir ? 0 : p.seq_len - 1);
  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? step : (p.seq_len - 1 - step);
    const int toff = t * stride_i;
    const bool first_ = dir ? (t == p.seq_len - 1) : (t == 0);          // first step of the forward recurrence: c_{-1} = 0
    const bool last = step + 1 == p.seq_len;
    NST_(0);
    // ---- 1. gate gradients of the owned units: LDS tile (own columns) + the gates output (write-through: the partner reads them)
    if constexpr (NS_PFI && HELP == 0 && TCH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the prefetch's LDS-DMAs (the compiler does not count them): long landed
    if (active) {
      uint2 gpre[2][4];
      float cpre[2][4];
      bf16_t dhpre[2][4];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rowb(rt, r) + toff;
#ifdef NSABL_NO_LOAD      // timing diagnostics (wrong results): NSABL_NO_LOAD, NSABL_NO_STORE, NSABL_NO_MM, NSABL_NO_POLL, NSABL_NO_COPY (scripts/abl_nsplit.py)
          gpre[rt][r] = make_uint2((unsigned)row, 0x3f003f00u); cpre[rt][r] = (float)(row & 3); dhpre[rt][r] = (bf16_t)(0x3c00 + (row & 7));
#else
          if constexpr (NS_PFI && HELP == 0 && TCH == 0) {
            gpre[rt][r] = *reinterpret_cast<const uint2*>(pfg + (rt * 16 + lr * 4 + r) * 128 + lc * 8);      // (requested a product ago: see prefetch_inputs)
            cpre[rt][r] = cn[rt][r];
            dhpre[rt][r] = dhn[rt][r];
          } else {
            gpre[rt][r] = *reinterpret_cast<const uint2*>(gates + ((long)row * ldg_i + (gcol_i + uc * 4)));
            cpre[rt][r] = first_ ? 0.f : p.c[(long)(row + prev_i) * ldc_i + (hcol_i + uc)];
            dhpre[rt][r] = dh[(long)row * ldd_i + (hcol_i + uc)];
          }
#endif
        }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float iv = __uint_as_float(gpre[rt][r].x << 16), fv = __uint_as_float(gpre[rt][r].x & 0xffff0000u);
          const float gv = __uint_as_float(gpre[rt][r].y << 16), ov = __uint_as_float(gpre[rt][r].y & 0xffff0000u);
          const float dht = bf16_to_f32(dhpre[rt][r]) + dhr[rt][r];
          const float tc = tanhf_(ccur[rt][r]);
          const float dct = dcs[rt][r] + dht * ov * (1.f - tc * tc);
          const float d0 = dct * gv * iv * (1.f - iv), d1 = dct * cpre[rt][r] * fv * (1.f - fv);
          const float d2 = dct * iv * (1.f - gv * gv), d3 = dht * tc * ov * (1.f - ov);
          dcs[rt][r] = dct * fv;
          ccur[rt][r] = cpre[rt][r];                                     // c_{t-1} is the next processed step's c_t
          uint2 pk = make_uint2(0u, 0u);
          if (uvalid) {
            pk.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
            pk.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
          }
          if (uvalid) *reinterpret_cast<uint2*>(tile + (rt * 16 + lr * 4 + r) * PITCH + u * 8) = pk;     // (columns past 4H stay zero)
        }
    }
    // (raw barriers: __syncthreads() would also drain the write-through stores above - their latency belongs behind the weight stream)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NST_(1);
    __builtin_amdgcn_s_barrier();                                        // the own half of the tile is complete
    NST_(2);
    // the own half of the tile -> the gates output, 16 bytes per lane along the rows (write-through).  (Stored from the cell phase,
    // 8 bytes per lane and unit, the same bytes cost 3 us per step: an sc1 store of 8 bytes per lane moves at a third of the 16-byte
    // form's rate per byte, MI355X_MICROARCH.md "stores of each flavour"; profiles/r04_abl_nsplit_v1.log.)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int ob0 = m ? UT0 * 128 : 0, ocpr = ((m ? G4 * 2 : UT0 * 128) - ob0) / 16;       // own bytes [ob0, ob0 + 16 ocpr) of a row
    const unsigned step_off = (unsigned)((long)toff * ldg_i * 2);
    if constexpr (HELP == 0) {
#ifdef NSABL_NO_STORE
      for (int idx = tid; idx < 0 * ocpr; idx += NSTHR) {
#else
      for (int idx = tid; idx < 32 * ocpr; idx += NSTHR) {
#endif
        const int row = idx / ocpr, cc = idx - row * ocpr;
        const int grow = rowtab[row];
        if (grow < 0) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(tile + row * PITCH + ob0 + cc * 16);
        const unsigned off = (unsigned)(((long)(grow + toff) * ldg_i + gcol_i) * 2 + ob0 + cc * 16);
        if (local) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)off, 0, 0);       // stays in the pair's L2
        else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_g, (int)off, 0, 16);           // sc1: write-through
      }
    }
    NST_(3);
    if (last) break;                                                     // (the last step's gradients are stored; nothing waits for them)
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const char* ar = tile + lc * PITCH + 16 * lr;
    // k-slabs [k0, k1) of this wave's unit tile against the tile in LDS, NS_KB fragments in flight; `publish`: after the first batch
    // of fragment loads is issued, wait for this wave's gate-gradient stores (older in the in-order vmcnt queue) and count the wave in
    auto product = [&](auto kbc, int k0, int k1, bool publish) {
      constexpr int KB = decltype(kbc)::value;      // fragments in flight (the publishing product: NS_KB = 9, what its counted wait is written for)
#pragma unroll 1
      for (int kb = k0; kb < k1; kb += KB) {
        uint4 b[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int ks = (kb + i < k1) ? kb + i : k1 - 1;
          b[i] = *reinterpret_cast<const uint4*>(whhT + (long)ks * 1024);
        }
        if (publish && kb == k0 + NS_PUB * KB) {
          // this batch's NS_KB fragment loads may stay in flight; everything older - the earlier batches and, before them, this wave's
          // gate-gradient stores - is then complete.  (In front of the FIRST batch this wait put the write-through stores' ~2 us of
          // acknowledge latency on the critical path: profiles/r04_abl_nsplit_v1.log, 14.5 -> 11.6 us per step without the stores.)
          asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
          if (lane == 0) {
            const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
            if (n == (unsigned)NSW * (unsigned)(step + 1))               // every wave of this member has waited for its stores
              __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (sc1 store)
          }
        }
#ifndef NSABL_NO_MM
#pragma unroll
#else
#pragma unroll
        for (int i = 0; i < 1; ++i) acc[0][0] += __uint_as_float(b[0].x);
        if (false)
#endif
        for (int i = 0; i < KB; ++i) {
          if (kb + i < k1) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + (kb + i) * 64);
              acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b[i]), acc[rt], 0, 0, 0);
            }
          }
        }
      }
    };
    static_assert(NS_KB == 9, "the counted vmcnt above is written for 9 fragments in flight");
    if constexpr (HELP > 0) {
      // ---- 2 + 3 (helper form). compute waves: the own K range.  Helper waves: wait for the own half's stores, raise the flag, wait for the
      // partner's, copy its half into the tile.  One barrier ends both.
      if (active) {
        product(std::integral_constant<int, NS_KB>{}, ks_own0, ks_own1, false);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
    // ---- 2. the own K range (in LDS already); the hand-off to the partner travels meanwhile
    if (active) {
      product(std::integral_constant<int, NS_KB>{}, ks_own0, ks_own1, true);
    } else {                                                             // a wave without a unit tile stored its share of the rows too
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
        if (n == (unsigned)NSW * (unsigned)(step + 1)) __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    NST_(4);
    // ---- 3. the partner's half: wait for its flag, copy its columns of the 32 rows from the gates output into the tile
    if (w == 0) {
      unsigned spins = 0;
      if (lane == 0) {
#ifdef NSABL_NO_POLL
        while (false && __hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
#else
        while (!dead && __hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
#endif
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { dead = true; atomicExch(p.err, 1u); lsync[1] = 1u; }
        }
      }
    }
    __syncthreads();
    NST_(5);
#ifdef NSABL_NO_COPY
    for (int idx = tid; idx < 0 * pcpr; idx += NSTHR) {
#else
    for (int idx = tid; idx < 32 * pcpr; idx += NSTHR) {
#endif
      const int row = idx / pcpr, cc = idx - row * pcpr;
      const int grow = rowtab[row];
      if (grow < 0) continue;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const unsigned off = (unsigned)(((long)(grow + toff) * ldg_i + gcol_i) * 2 + pb0 + cc * 16);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)off, 0, 16);                 // sc1: L1-bypassing
      *reinterpret_cast<uint4*>(tile + row * PITCH + pb0 + cc * 16) = make_uint4(v[0], v[1], v[2], v[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }
    NST_(6);
    if constexpr (NS_PFI && HELP == 0 && TCH == 0) prefetch_inputs(dir ? t + 1 : t - 1);      // (this is not the last step: that one left the loop above)
    // ---- 4. the other K range
    if (active) {
      // the first NS_RES k-slabs of the partner's range come from LDS (resident for the whole launch), the rest is streamed
      const int k0o = m ? 0 : ks_own1, k1o = m ? ks_own0 : NSLAB;
#if NS_RES > 0
#ifndef NSABL_NO_MM
#pragma unroll
      for (int i = 0; i < NS_RES; ++i) {
        const uint4 bw = *reinterpret_cast<const uint4*>(resw + i * 1024);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + (k0o + i) * 64);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bw), acc[rt], 0, 0, 0);
        }
      }
#endif
#endif
      product(std::integral_constant<int, NS_KB2>{}, k0o + NS_RES, k1o, false);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[rt][r] = acc[rt][r];
    }
    NST_(7);
    // no barrier here: the next step's cell phase rewrites the OWN columns, whose last readers (this step's own-range product) sit
    // behind two barriers; the next copy rewrites the PARTNER's columns behind the next step's two barriers
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
#ifdef URSE_EXPERIMENTS      // measured losers of round 5: compiled only into variant builds (scripts/build_variant.sh <name> lstm_nsplit "-DURSE_EXPERIMENTS")
// Round 5 EXPERIMENT (opt-in, URSE_NSPLIT_WIDE=1): the same pair protocol with SEVEN waves of TWO unit tiles each (8-wave budget: 256 registers).
// The question: three kernels looked like one law - the 16-wave streaming BPTT keeps 16 x 13 KB of weight fragments in flight and streams at 110 GB/s,
// the 13-wave kernel above 13 x 9 KB at 62 GB/s, the three-member kernel (lstm_nsplit3.hip) 9 x 11 KB at 52 GB/s: bytes in flight / 1.9 us every time -
// is the weight stream bound by what the waves can keep IN FLIGHT, i.e. by registers?  This form keeps two streams of NSW_KB fragments per wave in
// flight (7 x 2 x 14 KB = 196 KB), reads one A fragment from LDS for both tiles, takes the cell-phase inputs through 32-bit buffer offsets and moves
// 16-byte pieces with a fixed row per thread.  ANSWER: no.  Bit-identical gate gradients, 5.77 ms against 5.08 per launch; 7, 10 or 14 fragments
// in flight per tile take the same time (profiles/r05_abl_nsplit_wide_v1.log); with everything but the weight stream switched off the step is
// 9.4 us = 640 KB at 68 GB/s - the rate at which ONE CU reads its L2, whatever is in flight.  The N-split is bound by bytes through the CU.
// Second question (the stamps of the 13-wave kernel put 30 % of a step on the exposed round trip of the input rows): with 256 registers the NEXT step's
// inputs fit beside the weights - requested behind this step's last weight fragments (NSW_PFX; in front of them they would hold every younger load back in
// the in-order vmcnt queue).  ANSWER: bit-identical, 5.93 ms - no gain: "behind the last fragments" is 300 cycles before the cell phase, not a step.
constexpr int NSW_W = 7;            // waves per workgroup
constexpr int NSW_THR = NSW_W * 64;
#ifndef NSW_KB
#define NSW_KB 5                    // weight fragments in flight per wave AND tile (7, 10, 14 measured equal; 5 leaves the registers the prefetched inputs need)
#endif
#ifndef NSW_PFX
#define NSW_PFX 1                   // the next step's input rows requested a step ahead
#endif

template <int H>
__global__ void __launch_bounds__(NSW_THR) lstm_bwd_nsplitw_kernel(NsplitArgs p) {
  constexpr int NUT = (H + 15) / 16, G4 = 4 * H, NSLAB = G4 * 2 / 64, UT0 = (NUT + 1) / 2;     // 25 tiles, 49 slabs, member 0 owns 13 tiles
  static_assert(UT0 <= 2 * NSW_W, "two unit tiles per wave");
  constexpr int PITCH = lds_frag_pitch(G4 * 2);
  constexpr int TPR = NSW_THR / 32;                                    // 14 threads per row move the row's 16-byte pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* tile = smem;                                                   // [32][PITCH] gate gradients of the step, all units (MFMA A operand)
  unsigned* lsync = reinterpret_cast<unsigned*>(smem + 32 * PITCH);    // [0] waves whose stores are complete (monotonic), [1] dead flag, [2] same XCD
  int* rowtab = reinterpret_cast<int*>(lsync + 4);
  const int tid = threadIdx.x, lane = tid & 63, lr = lane >> 4, lc = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = blockIdx.x, m = (lin >> 3) & 1, P = (lin >> 4) * 8 + (lin & 7);
  const int dir = P & 1, pair = P >> 1;
  if (pair >= p.npairs) return;
  const int ut_lo = m ? UT0 : 0, ut_hi = m ? NUT : UT0;                 // owned unit tiles
  const int ut0 = ut_lo + 2 * w;                                        // this wave's tiles: ut0, ut0 + 1
  const int ntile = ut0 + 1 < ut_hi ? 2 : (ut0 < ut_hi ? 1 : 0);
  const int ob0 = ut_lo * 128, ob1 = (ut_hi * 128 < G4 * 2) ? ut_hi * 128 : G4 * 2;
  const int ks_own0 = 2 * ut_lo, ks_own1 = (2 * ut_hi < NSLAB) ? 2 * ut_hi : NSLAB;
  int ucol[2];
  bool uval[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int u = (ut0 + tt) * 16 + lc;
    uval[tt] = tt < ntile && u < H;
    ucol[tt] = uval[tt] ? u : H - 1;
  }
  if (tid < 3) lsync[tid] = 0u;
  const int s0 = pair * 32;
  const int ldg_i = (int)p.ldg, ldd_i = (int)p.ldd, ldc_i = 2 * H, stride_i = (int)p.stride;
  const int gcol_i = dir * G4, hcol_i = dir * H, prev_i = dir ? stride_i : -stride_i;
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(p.gates, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dh), 0, (int)p.d_bytes, 0x00020000);
  unsigned rowq[2][4];                                                  // row of (sequence, t = 0) of the lane's rows in the C layout (clamped beyond n_seq)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int seq = s0 + rt * 16 + lr * 4 + r;
      if (seq >= p.n_seq) seq = p.n_seq - 1;
      rowq[rt][r] = (unsigned)((seq / p.inner) * p.outer + (seq % p.inner));
    }
  const unsigned ldg2 = (unsigned)ldg_i * 2u, ldc4 = (unsigned)ldc_i * 4u, ldd2 = (unsigned)ldd_i * 2u;      // (< 2^24, as the rows: checked by the host)
  const char* whhT0 = reinterpret_cast<const char*>(p.whhT) + ((long)dir * NUT * NSLAB + (long)(ntile > 0 ? ut0 : ut_lo) * NSLAB) * 1024 + lane * 16;
  const char* whhT1 = whhT0 + (ntile > 1 ? (long)NSLAB * 1024 : 0);
  unsigned* my_flag = p.flags + ((dir * p.npairs + pair) * 2 + m);
  unsigned* partner_flag = p.flags + ((dir * p.npairs + pair) * 2 + (m ^ 1));
  if (tid < 32) {
    const int seq = s0 + tid;
    rowtab[tid] = seq < p.n_seq ? (int)((seq / p.inner) * p.outer + (seq % p.inner)) : -1;
  }
  for (int i = tid; i < 32 * PITCH / 16; i += NSW_THR) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);

  float dcs[2][2][4], dhr[2][2][4], ccur[2][2][4];
  {
    const int toff0 = (dir ? 0 : p.seq_len - 1) * stride_i;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dcs[tt][rt][r] = 0.f;
          dhr[tt][rt][r] = 0.f;
          ccur[tt][rt][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(__umul24(rowq[rt][r], ldc4) + (unsigned)(hcol_i + ucol[tt]) * 4u), toff0 * ldc_i * 4, 0));
        }
  }
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;      // HW_REG_XCC_ID
    unsigned* xw = p.flags + 2 * 2 * p.npairs + ((dir * p.npairs + pair) * 2);
    __hip_atomic_store(xw + m, xcc | 0x100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v = 0u, spins = 0;
    while ((v = __hip_atomic_load(xw + (m ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }
    }
    lsync[2] = (v == (xcc | 0x100u)) ? 1u : 0u;
  }
  __syncthreads();
  const bool local = lsync[2] != 0u;
  bool dead = false;
  // the thread's row of the tile for the 16-byte pieces it moves (own half out, the partner's half in): row tid / 14, pieces tid % 14 + 14 j
  const int mv_row = tid / TPR, mv_c0 = tid - mv_row * TPR;
  const int mv_grow = rowtab[mv_row];
  const unsigned mv_lds = (unsigned)(mv_row * PITCH + mv_c0 * 16);
  const unsigned mv_glb = mv_grow >= 0 ? (unsigned)(((long)mv_grow * ldg_i + gcol_i) * 2 + mv_c0 * 16) : 0xFFFFF000u;      // (a row past n_seq: loads return zeros, stores are dropped)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int MAXP = (UT0 * 128 / 16 + TPR - 1) / TPR;                // 8 rounds of 14 pieces cover the wider half (104 pieces)
  const int pb0 = m ? 0 : UT0 * 128, pb1 = m ? UT0 * 128 : G4 * 2;      // the partner's bytes of a row

  // the step's input rows (saved gate activations, c_{t-1}, dh) of the lane's 2 tiles x 2 x 4 rows: ALL requested at once (one round trip, not one per
  // tile and row group), and with NSW_PFX a step ahead - behind the last weight fragments of the previous step, where they are the youngest entries of the
  // in-order vmcnt queue and hold nothing back; the 13-wave kernel has no registers for this (124 of 128), this one has
  uint2 gpre[2][2][4];
  float cpre[2][2][4];
  bf16_t dhpre[2][2][4];
  auto load_inputs = [&](int toff_, bool first__) __attribute__((always_inline)) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const unsigned gbase = (unsigned)(gcol_i + ucol[tt] * 4) * 2u, cbase = (unsigned)(hcol_i + ucol[tt]) * 4u, dbase = (unsigned)(hcol_i + ucol[tt]) * 2u;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef NSABL_NO_LOAD
          gpre[tt][rt][r] = make_uint2(rowq[rt][r], 0x3f003f00u); cpre[tt][rt][r] = (float)(toff_ & 3); dhpre[tt][rt][r] = (bf16_t)(0x3c00 + (toff_ & 7));
#else
          if (tt < ntile) {
            const u32x2 gv2 = __builtin_amdgcn_raw_buffer_load_b64(rs_g, (int)(__umul24(rowq[rt][r], ldg2) + gbase), toff_ * ldg_i * 2, 0);
            gpre[tt][rt][r] = make_uint2(gv2[0], gv2[1]);
            cpre[tt][rt][r] = first__ ? 0.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, (int)(__umul24(rowq[rt][r], ldc4) + cbase), (toff_ + prev_i) * ldc_i * 4, 0));
            dhpre[tt][rt][r] = (bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_d, (int)(__umul24(rowq[rt][r], ldd2) + dbase), toff_ * ldd_i * 2, 0);
          }
#endif
        }
    }
  };
  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? step : (p.seq_len - 1 - step);
    const int toff = t * stride_i;
    const bool first_ = dir ? (t == p.seq_len - 1) : (t == 0);          // first step of the forward recurrence: c_{-1} = 0
    const bool last = step + 1 == p.seq_len;
    // (the row registers are made opaque per step: visible as loop invariants, the offsets derived from them are hoisted out of the time loop and spilled)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) asm volatile("" : "+v"(rowq[rt][0]), "+v"(rowq[rt][1]), "+v"(rowq[rt][2]), "+v"(rowq[rt][3]));
    // ---- 1. gate gradients of the owned units -> LDS tile (own columns)
#if NSW_PFX == 0
    load_inputs(toff, first_);
#else
    if (step == 0) load_inputs(toff, first_);                            // (later steps: requested behind the last weight fragments of the previous step)
#endif
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      if (tt < ntile) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint2 gp = gpre[tt][rt][r];
            const float cp = cpre[tt][rt][r];
            const float iv = __uint_as_float(gp.x << 16), fv = __uint_as_float(gp.x & 0xffff0000u);
            const float gv = __uint_as_float(gp.y << 16), ov = __uint_as_float(gp.y & 0xffff0000u);
            const float dht = bf16_to_f32(dhpre[tt][rt][r]) + dhr[tt][rt][r];
            const float tc = tanhf_(ccur[tt][rt][r]);
            const float dct = dcs[tt][rt][r] + dht * ov * (1.f - tc * tc);
            const float d0 = dct * gv * iv * (1.f - iv), d1 = dct * cp * fv * (1.f - fv);
            const float d2 = dct * iv * (1.f - gv * gv), d3 = dht * tc * ov * (1.f - ov);
            dcs[tt][rt][r] = dct * fv;
            ccur[tt][rt][r] = cp;                                        // c_{t-1} is the next processed step's c_t
            if (uval[tt]) {
              uint2 pk;
              pk.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
              pk.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
              *reinterpret_cast<uint2*>(tile + (rt * 16 + lr * 4 + r) * PITCH + ((ut0 + tt) * 16 + lc) * 8) = pk;     // (columns past 4H stay zero)
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // the own half of the tile is complete
    // the own half of the tile -> the gates output, 16 bytes per lane along the rows
    const unsigned step_off = (unsigned)((long)toff * ldg_i * 2);
#ifndef NSABL_NO_STORE
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
    if (last) break;
    f32x4_t acc[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[tt][rt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const char* ar = tile + lc * PITCH + 16 * lr;
    // k-slabs [k0, k1) of this wave's unit tiles against the tile in LDS: NSW_KB fragments of EACH tile in flight; `publish`: behind the second batch of
    // fragment loads wait for this wave's stores (older in the in-order vmcnt queue) and count the wave in
    auto product = [&](int k0, int k1, bool publish) {
#pragma unroll 1
      for (int kb = k0; kb < k1; kb += NSW_KB) {
        uint4 b0[NSW_KB], b1[NSW_KB];
#pragma unroll
        for (int i = 0; i < NSW_KB; ++i) {
          const int ks = (kb + i < k1) ? kb + i : k1 - 1;
          b0[i] = *reinterpret_cast<const uint4*>(whhT0 + (long)ks * 1024);
          b1[i] = *reinterpret_cast<const uint4*>(whhT1 + (long)ks * 1024);
        }
        if (publish && (kb == k0 + NSW_KB || (kb == k0 && k0 + NSW_KB >= k1))) {
          asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NSW_KB) : "memory");
          if (lane == 0) {
            const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
            if (n == (unsigned)NSW_W * (unsigned)(step + 1))             // every wave of this member has waited for its stores
              __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#ifndef NSABL_NO_MM
#pragma unroll
        for (int i = 0; i < NSW_KB; ++i) {
          if (kb + i < k1) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + (kb + i) * 64);
              acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b0[i]), acc[0][rt], 0, 0, 0);
              acc[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b1[i]), acc[1][rt], 0, 0, 0);
            }
          }
        }
#else
        acc[0][0][0] += __uint_as_float(b0[0].x ^ b1[0].x);
#endif
      }
    };
    static_assert(2 * NSW_KB <= 63, "counted vmcnt");
    // ---- 2. the own K range (in LDS already); the hand-off to the partner travels meanwhile
    if (ntile > 0) {
      product(ks_own0, ks_own1, true);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) {
        const unsigned n = __hip_atomic_fetch_add(&lsync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
        if (n == (unsigned)NSW_W * (unsigned)(step + 1)) __hip_atomic_store(my_flag, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- 3. the partner's half: wait for its flag, copy its columns of the 32 rows from the gates output into the tile
    if (w == 0 && lane == 0) {
      unsigned spins = 0;
#ifndef NSABL_NO_POLL
      while (!dead && __hip_atomic_load(partner_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { dead = true; atomicExch(p.err, 1u); lsync[1] = 1u; }
      }
#endif
    }
    __syncthreads();
#ifndef NSABL_NO_COPY
    {
      u32x4 v[MAXP];
#pragma unroll
      for (int jj = 0; jj < MAXP; ++jj) {
        const int cb = pb0 + (mv_c0 + TPR * jj) * 16;
        v[jj] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (int)(cb < pb1 ? mv_glb + (unsigned)(pb0 + TPR * 16 * jj) : 0xFFFFF000u), (int)step_off, 16);      // sc1: L1-bypassing
      }
#pragma unroll
      for (int jj = 0; jj < MAXP; ++jj) {
        const int cb = pb0 + (mv_c0 + TPR * jj) * 16;
        if (cb < pb1) *reinterpret_cast<uint4*>(tile + mv_lds + pb0 + TPR * 16 * jj) = make_uint4(v[jj][0], v[jj][1], v[jj][2], v[jj][3]);
      }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- 4. the other K range
    if (ntile > 0) {
      const int k0o = m ? 0 : ks_own1, k1o = m ? ks_own0 : NSLAB;
#if NSW_PFX > 0
      // all batches but the last in the loop; the last one written out, so that the NEXT step's input rows can be requested between its weight loads and
      // its MFMAs in straight-line code (inside the loop, behind a condition, the 64 input registers doubled through the loop's merges)
      const int klast = k0o + ((k1o - k0o - 1) / NSW_KB) * NSW_KB;
      product(k0o, klast, false);
      {
        uint4 b0[NSW_KB], b1[NSW_KB];
#pragma unroll
        for (int i = 0; i < NSW_KB; ++i) {
          const int ks = (klast + i < k1o) ? klast + i : k1o - 1;
          b0[i] = *reinterpret_cast<const uint4*>(whhT0 + (long)ks * 1024);
          b1[i] = *reinterpret_cast<const uint4*>(whhT1 + (long)ks * 1024);
        }
        {
          const int tn = dir ? step + 1 : (p.seq_len - 2 - step);
          load_inputs(tn * stride_i, dir ? (tn == p.seq_len - 1) : (tn == 0));
        }
#ifndef NSABL_NO_MM
#pragma unroll
        for (int i = 0; i < NSW_KB; ++i) {
          if (klast + i < k1o) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
              const uint4 a = *reinterpret_cast<const uint4*>(ar + rt * 16 * PITCH + (klast + i) * 64);
              acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b0[i]), acc[0][rt], 0, 0, 0);
              acc[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b1[i]), acc[1][rt], 0, 0, 0);
            }
          }
        }
#endif
      }
#else
      product(k0o, k1o, false);
#endif
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) dhr[tt][rt][r] = acc[tt][rt][r];
    }
  }
}

#endif   // URSE_EXPERIMENTS

}  // namespace urse

using namespace urse;

// -> plan {pairs per direction, workgroups, flag words}; < 0 (URSE_ERR_UNSUPPORTED) if the shape has no kernel or the pairs would not
// be co-resident beside the reserved CUs
#ifdef NSSTAMP
extern "C" int urse_diag_nsplit_stamps(void* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nsstamps), sizeof(unsigned long long) * 512 * 16);
}
#endif

extern "C" int urse_lstm_nsplit_plan(int H, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && n_seq > 0 && reserved_cus >= 0, "urse_lstm_nsplit_plan: bad argument");
  if (H != 392) {
    set_error("urse_lstm_nsplit_plan: unsupported H=%d", H);
    return URSE_ERR_UNSUPPORTED;
  }
  const int npairs = (n_seq + 31) / 32;
  const int wgs = ((2 * npairs + 7) / 8) * 16;
  if (wgs > device_cu_count() - reserved_cus) {
    set_error("urse_lstm_nsplit_plan: %d workgroups do not fit beside %d reserved CUs", wgs, reserved_cus);
    return URSE_ERR_UNSUPPORTED;
  }
  plan[0] = npairs; plan[1] = wgs; plan[2] = 2L * (2L * npairs * 2);      // step flags + XCC ids
  return URSE_OK;
}

extern "C" int urse_lstm_nsplit_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT, void* flags,
                                    void* err_flag, int H, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                                    int reserved_cus, void* stream) {
  URSE_CHECK_ARG(dh && gates && c && whhT && flags && err_flag, "urse_lstm_nsplit_bwd: null pointer");
  int64_t plan[3];
  int rc = urse_lstm_nsplit_plan(H, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(seq_len > 0 && inner > 0 && ldg >= 8L * H && ldd >= 2L * H && ldg % 8 == 0 && ((uintptr_t)gates % 16) == 0,
                 "urse_lstm_nsplit_bwd: bad leading dimension / alignment");
  const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
  URSE_CHECK_ARG(rows * ldg * 2 < 0xFFFFF000L && ldd < (1L << 31), "urse_lstm_nsplit_bwd: the gates matrix exceeds 32-bit byte offsets");
  NsplitArgs p;
  p.dh = dh; p.ldd = ldd; p.gates = gates; p.ldg = ldg; p.c = c; p.whhT = whhT; p.flags = (unsigned*)flags; p.err = (unsigned*)err_flag;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len; p.npairs = (int)plan[0];
  p.g_bytes = (unsigned)(rows * ldg * 2); p.c_bytes = (unsigned)(rows * 2L * H * 4); p.d_bytes = (unsigned)(rows * ldd * 2);
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(flags, 0, sizeof(unsigned) * plan[2], st);
  const size_t lds = (size_t)32 * lds_frag_pitch(4 * 392 * 2) + 256 + (size_t)NSW * (NS_PFI ? 4096 : NS_RES * 1024);      // tile, flags + row table, the prefetched gates (or the resident fragments)
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_nsplit_kernel<392, 0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  note_launch(URSE_KV_LSTM_BWD_NSPLIT);
#ifdef URSE_EXPERIMENTS
  // Variant builds only (round 6: the shipped library instantiates the 13-wave form alone).  What each measured, all in DESIGN 9.6:
  // URSE_NSPLIT_HELPERS=3, three helper waves own the hand-off: 0.07 ms faster alone, no faster in the step (133.09 / 130.96 with, 131.70 / 131.63 without,
  // profiles/r05_ab_nsplit_helpers_v3.log); URSE_NSPLIT_WIDE=1, seven waves of two unit tiles: 5.77 vs 5.08 ms; URSE_NSPLIT_TOUCH=1, a wave that warms the
  // next step's input sectors in L2: 6.41 vs 5.07 ms.
  static bool once_x = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_nsplit_kernel<392, 3>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_nsplitw_kernel<392>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_nsplit_kernel<392, 0, 1>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once_x;
  const int helpers = getenv("URSE_NSPLIT_HELPERS") ? atoi(getenv("URSE_NSPLIT_HELPERS")) : 0;
  const int wide = getenv("URSE_NSPLIT_WIDE") ? atoi(getenv("URSE_NSPLIT_WIDE")) : 0;
  const bool wide_ok = wide && rows < (1L << 24) && ldg * 2 < (1L << 24) && ldd * 2 < (1L << 24) && rows * 2L * H * 4 < 0xFFFFF000L && rows * ldd * 2 < 0xFFFFF000L;
  const int touch = getenv("URSE_NSPLIT_TOUCH") ? atoi(getenv("URSE_NSPLIT_TOUCH")) : 0;
  if (wide_ok) hipLaunchKernelGGL((lstm_bwd_nsplitw_kernel<392>), dim3((unsigned)plan[1]), dim3(NSW_THR), lds, st, p);
  else if (touch > 0 && helpers == 0) hipLaunchKernelGGL((lstm_bwd_nsplit_kernel<392, 0, 1>), dim3((unsigned)plan[1]), dim3(NSTHR + 64), lds, st, p);
  else if (helpers > 0) hipLaunchKernelGGL((lstm_bwd_nsplit_kernel<392, 3>), dim3((unsigned)plan[1]), dim3((NSW + 3) * 64), lds, st, p);
  else
#endif
  hipLaunchKernelGGL((lstm_bwd_nsplit_kernel<392, 0>), dim3((unsigned)plan[1]), dim3(NSTHR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_nsplit_bwd");
  return URSE_OK;
}
