// Persistent "cluster" LSTM forward recurrence: recurrent weights RESIDENT IN REGISTERS for the whole sequence.
//
// lstm.hip streams W_hh (1.2 MB bf16 per direction) from L2 on every time step and is bound by the ~70 GB/s at which
// one CU can read its L2 (18 us per step).  Here a cluster of C workgroups (one per CU) shares a set of sequences:
// workgroup j owns 14 "quads" of hidden units (4 units x 4 gates = one 16-column MFMA B tile per quad, one quad per
// wave), keeps those B fragments in VGPRs (13 x 16 B per lane for Hp = 416), and per step
//   1. loads the cluster's h_{t-1} rows [64, Hp] from the exchange buffer into LDS (MFMA A operand), polling every
//      16-byte chunk until the step tag embedded in its elements is current (see the protocol note in the kernel),
//   2. every wave computes its quad's 16 gate columns for the 64 rows, transposes the accumulator inside each lane
//      quad with DPP so that one lane holds i,f,g,o of one (row, unit), applies the LSTM cell (c_t stays in registers),
//   3. stages h_t through LDS and writes it with 16-byte stores to the exchange buffer (sc1, tagged) and to hout.
// Placement-independent: visibility comes from sc1 (write-through) stores and sc1 (L1-bypassing) loads of data that
// validates itself; spins are bounded (error flag); grid <= 256 workgroups so all are co-resident.
// Same math / layouts as lstm.hip (gate-interleaved gx, bf16 h, f32 c); bf16 only (the f32 parity mode keeps lstm.hip).
#include "urse_common.h"

namespace urse {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int CW = 14;            // working waves per workgroup (one unit quad each)
constexpr int CTHR = CW * 64;     // 896 threads
constexpr int CROWS = 64;         // rows (sequences) per chunk = 4 MFMA row tiles
constexpr int UW = CW * 4;        // hidden units per workgroup (56)

struct ClusterArgs {
  void* gx; long ldg;
  const void* whhq;               // [2][nq][NSLAB][64][16 B] quad-ordered fragments
  void* hout; long ldh;
  void* hout2;                    // f16 operands: h once more in bf16 for the weight-gradient GEMMs (null: not wanted)
  float* c;
  bf16_t* hx;                     // exchange [2 parity][2 dir][ncl][rows_pad][Hp]
  unsigned* cnt;                  // [2 dir][ncl] arrival counters (zeroed per launch)
  unsigned* err;                  // timeout flag
  int H, Hp, save;
  long inner, outer, stride;
  int n_seq, seq_len;
  int C, ncl, rows_per_cluster, rows_pad;
  unsigned g_bytes, c_bytes, h_bytes;   // sizes of the gx / c / hout matrices (range-checked deferred stores)
  unsigned* xws;                  // XCD-aware formation (null = static clusters): [0..7] arrivals per XCD, [8] arrivals, zeroed per launch
};

__device__ __forceinline__ float quad_bcast(float v, int k) {
  // value of quad-lane k, broadcast inside each group of 4 lanes (DPP quad_perm)
  int r;
  const int iv = __float_as_int(v);
  switch (k) {
    case 0: r = __builtin_amdgcn_mov_dpp(iv, 0x00, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_mov_dpp(iv, 0x55, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_mov_dpp(iv, 0xAA, 0xf, 0xf, true); break;
    default: r = __builtin_amdgcn_mov_dpp(iv, 0xFF, 0xf, 0xf, true); break;
  }
  return __int_as_float(r);
}

// write-through (sc1) 16-byte accesses to the exchange buffer: bypass this CU's L1 on loads, leave L2 on stores, so the
// hand-off needs no release / acquire fence (MI355X_MICROARCH "Valid forms": every payload store and load sc1, every
// storing wave drains vmcnt, one lane per workgroup signals with an agent-scope atomic, the poller is an sc1 load).
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 16);
}
__device__ __forceinline__ void store_plain(__amdgpu_buffer_rsrc_t rs, unsigned off, uint4 v) {      // keeps the line in this XCD's L2
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs, (int)off, 0, 0);
}
__device__ __forceinline__ uint4 load_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// HELP > 0: that many HELPER waves beside the 14 working waves.  They own everything that is not on the recurrence's critical path: the
// previous step's plain stores (h_t -> hout, saved gates, c_t: 3,136 16-byte pieces) and the next step's gate pre-activations (1,792
// pieces), between the same two barriers as the working waves, whose instruction streams and in-order vmcnt queues then hold the h gather,
// the MFMAs, the cell update and the publication only.  (The ablation of the helper-less form priced the deferred stores at 1.7 us and the
// staged pre-activations at 1.3 us of a 6.4 us step - at 21 GB/s per CU this kernel is bound by its step latency, not by bytes.)
// TI: operand format (bf16_t | f16_t): the gate pre-activations read from gx, the resident W_hh fragments, the exchanged h and hout; the saved
// gate activations written back into gx are bf16 in both (they feed the BPTT).  The tag bit of the hand-off (bit 14 = the exponent's MSB) is
// clear for |h| <= 1 in either format.  H2 (f16 only): hout2 receives h in bf16 as well.
template <int NSLAB, int MAXCH, int HELP = 0, typename TI = bf16_t, bool H2 = false>
__global__ void __launch_bounds__(CTHR + 64 * HELP) lstm_fwd_cluster_kernel(ClusterArgs p) {
  static_assert(!H2 || __is_same(TI, f16_t), "the bf16 copy of h exists in the f16 mode only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  const int H = p.H;
  constexpr int Hp = NSLAB * 32, pitch = lds_frag_pitch(Hp * 2);      // compile-time: the index arithmetic below folds to shifts / multiplies
  char* htile = smem;                                    // [CROWS][pitch]
  bf16_t* hstage = reinterpret_cast<bf16_t*>(smem + CROWS * pitch);   // [CROWS][UW]
  // ---- which cluster, which member, which sequences -------------------------------------------------------------------
  // Static form: consecutive blockIdx.x form a cluster - under the round-robin dealing of workgroups to XCDs its seven members
  // sit on seven DIFFERENT XCDs and every byte of the h all-gather crosses the fabric (14.3 GB per launch against 7.5 algorithmic,
  // profiles/r02_pmc_hbm_traffic_v1.json).  XCD-aware form (p.xws): every workgroup registers under the XCC id it READS from the
  // hardware (never inferred from blockIdx), waits until the whole grid has registered, and the final per-XCD counts then assign -
  // identically in every workgroup - the first 7 * floor(n_x / 7) registrants of XCD x to clusters of their own XCD ("local":
  // h is published with plain stores that stay in that XCD's L2, MI355X_MICROARCH.md: 104-122 GB/s same-XCD against 62-70) and
  // the rest to mixed clusters (sc1 stores as before), which get fewer sequences.  A count pattern this cannot serve (odd cluster
  // counts per kind) falls back to the static form.  Correctness never depends on where a workgroup runs: a local cluster is local
  // because its members read the same XCC id.
  int dir = blockIdx.y, cl = blockIdx.x / p.C, j = blockIdx.x - cl * p.C;
  int seq0 = cl * p.rows_per_cluster, nrows_x = p.rows_per_cluster, clx = dir * p.ncl + cl;     // clx: index of the cluster's exchange block
  bool local = false;
  if (p.xws != nullptr) {
    int* xs = reinterpret_cast<int*>(smem + CROWS * pitch + CROWS * UW * 2 + 16);      // [12] broadcast of the assignment
    if (tid == 0) {
      const int grid = gridDim.x * gridDim.y;
      const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 7u;      // HW_REG_XCC_ID
      const unsigned rank = __hip_atomic_fetch_add(p.xws + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p.xws + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      bool ok = true;
      while (__hip_atomic_load(p.xws + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)grid) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) { ok = false; atomicExch(p.err, 1u); break; }
      }
      int n[8], S = 0, before_full = 0, before_left = 0;
      for (int x = 0; x < 8; ++x) {
        n[x] = (int)__hip_atomic_load(p.xws + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int f = n[x] / p.C;
        if (x < (int)xcc) { before_full += f; before_left += n[x] - f * p.C; }
        S += f;
      }
      const int NC = grid / p.C, Mx = NC - S;                        // clusters in all, mixed ones
      const int f_me = n[xcc] / p.C;
      int mode = 0;                                                  // 0 = static fallback, 1 = XCD-aware
      if (ok && S > 0 && (S & 1) == 0 && (Mx & 1) == 0) {
        const int ns = S / 2, nm = Mx / 2;                           // per direction
        int rows_m = 0;
        if (nm > 0 && CROWS * ns < p.n_seq) rows_m = (p.n_seq - CROWS * ns + nm - 1) / nm;
        const int rows_s = (p.n_seq - rows_m * nm + ns - 1) / ns;
        if (rows_m <= CROWS && rows_s <= CROWS && rows_s > 0) {
          mode = 1;
          int ci, jj, lc_;
          if ((int)rank < f_me * p.C) { ci = before_full + (int)rank / p.C; jj = (int)rank % p.C; lc_ = 1; }
          else { const int li = before_left + ((int)rank - f_me * p.C); ci = S + li / p.C; jj = li % p.C; lc_ = 0; }
          const int d = ci & 1, k = lc_ ? (ci >> 1) : ((ci - S) >> 1);            // direction, index among its kind in that direction
          const int s0 = lc_ ? k * rows_s : ns * rows_s + k * rows_m;
          int nr = lc_ ? rows_s : rows_m;
          if (s0 + nr > p.n_seq) nr = p.n_seq - s0 > 0 ? p.n_seq - s0 : 0;
          xs[0] = 1; xs[1] = d; xs[2] = ci; xs[3] = jj; xs[4] = s0 < p.n_seq ? s0 : 0; xs[5] = nr; xs[6] = lc_;
        }
      }
      if (!mode) xs[0] = 0;
      if (!ok) xs[0] = -1;
    }
    __syncthreads();
    if (xs[0] < 0) return;                                           // the grid never assembled: flagged, nothing written
    if (xs[0] == 1) { dir = xs[1]; clx = xs[2]; j = xs[3]; seq0 = xs[4]; nrows_x = xs[5]; local = xs[6] != 0; cl = clx; }
    __syncthreads();
  }
  const int nq = (H + 3) >> 2;
  const int qd = j * CW + w;                             // this wave's unit quad
  const bool qvalid = qd < nq;
  // The recurrent product runs with the operands swapped: A = the resident W_hh fragment (rows = the quad's 16 gate columns), B = the
  // h fragment (columns = sequences).  The register contents are the same either way (both operand layouts are "index lane % 16, 8 k's
  // by lane / 16"), but the accumulator then holds, in lane (lr, lc), the FOUR GATES acc[0..3] of unit lr of the quad for sequence lc
  // of the row tile - the cell update needs no cross-lane traffic (it used to start with a 4 x 4 transpose inside each lane quad:
  // 16 DPP moves + 12 selects per row tile, 112 of the ~500 vector instructions of a wave's step).
  const int ul = lr, rl = lc;                            // unit within the quad / sequence within the row tile
  const int u = qd * 4 + ul;
  const bool uvalid = qvalid && u < H;
  const int uc = uvalid ? u : H - 1;

  uint4 breg[NSLAB];                                     // resident B fragments
  {
    const char* src = reinterpret_cast<const char*>(p.whhq) + (((long)dir * nq + (qvalid ? qd : 0)) * NSLAB) * 1024 + lane * 16;
#pragma unroll
    for (int ks = 0; ks < NSLAB; ++ks) breg[ks] = *reinterpret_cast<const uint4*>(src + ks * 1024);
  }
  for (int i = tid; i < CROWS * UW / 2; i += CTHR) reinterpret_cast<unsigned*>(hstage)[i] = 0u;   // pad units stay 0
  for (int i = tid; i < CROWS * UW / 2; i += CTHR)
    reinterpret_cast<unsigned*>(smem + CROWS * pitch + CROWS * UW * 2 + 16 + 64 + CROWS * sizeof(int))[i] = 0u;   // (the second h staging tile)
  float cst[MAXCH][4];
#pragma unroll
  for (int a = 0; a < MAXCH; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) cst[a][b] = 0.f;

  int seq1 = seq0 + nrows_x;
  if (seq1 > p.n_seq) seq1 = p.n_seq;
  const int nrows = seq1 > seq0 ? seq1 - seq0 : 0;
  bf16_t* gx = reinterpret_cast<bf16_t*>(p.gx);
  bf16_t* hout = reinterpret_cast<bf16_t*>(p.hout);
  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * p.rows_pad * Hp * 2);
  const unsigned cl_bytes = (unsigned)((long)clx * p.rows_pad * Hp * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.hx, 0, (int)(2u * plane_bytes), 0x00020000);
  constexpr int cpr = Hp * 2 / 16;                       // 16-B chunks per h row
  constexpr int HL = (CROWS * 52 + CTHR - 1) / CTHR;     // h-tile chunks per thread (Hp <= 416)

  // row bookkeeping of this lane (row tile rt: sequence rt*16 + rl), hoisted out of the time loop: the run-time
  // divisions by `inner` cost more than the cell math of a step
  int rowb[4];                                            // 32-bit row indices: one v_mad_i64_i32 per address
  bool rowv[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int lrow = rt * 16 + rl;
    rowv[rt] = lrow < nrows;
    int seq = seq0 + lrow;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    rowb[rt] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }
  const int ldg_i = (int)p.ldg, ldh_i = (int)p.ldh, ldc_i = 2 * H, stride_i = (int)p.stride, gcol_i = dir * 4 * H, hcol_i = dir * H;
  // the one (row, 16-byte piece) of the staged h tile this thread publishes / stores per step
  constexpr int SCx = UW * 2 / 16;
  const int st_row = tid / SCx, st_cc = tid - st_row * SCx;
  int st_grow = 0;
  {
    int seq = seq0 + st_row;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    st_grow = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }

  unsigned* deadflag = reinterpret_cast<unsigned*>(smem + CROWS * pitch + CROWS * UW * 2);
  if (tid == 0) *deadflag = 0u;
  const int hchunks = H / 8;                             // 16-byte chunks of a row that carry data (H % 8 == 0)
  constexpr unsigned TAGM = 0x40004000u;                 // bit 14 of both bf16 halves: always 0 in |h| <= 1

  // Hand-off protocol ("tag in data"): every bf16 h element published through the exchange buffer carries, in its
  // never-used exponent MSB, one bit that toggles each time its parity plane is rewritten (the host zeroes the planes
  // before the launch, the first write carries 1).  A consumer simply re-loads (sc1, L1-bypassing) a 16-byte chunk
  // until all eight tags show the value expected for that step: no arrival counter, no store drain + barrier + atomic +
  // poll round trips on the critical path, and correct at any placement / at 2-byte store granularity.  A plane is
  // overwritten only by a producer that has already consumed every workgroup's h of the step in between, which those
  // workgroups published only after consuming the data being overwritten.
  // The step's plain stores (saved gates, c_t, h_t -> hout) are DEFERRED into the next step, behind the first round of its h gather:
  // issued at the end of their own step they sit in front of the gather's loads in the wave's in-order vmcnt queue, and the wait for
  // the first h chunk then also waits for their write acknowledges (~1.5 us per step); issued behind the gather's loads they complete
  // while the MFMAs run.  With them gone from the step's tail the third barrier goes too: the h tile is rewritten by the gather behind
  // barrier 2 of the step that read it, the staging tile by the cell phase behind the next step's barrier 1, which its last readers
  // (the publication and the deferred hout store) precede.
  // (they wait in LDS staging tiles - registers are at the 128 cap of 14 waves - and leave as 16-byte pieces along the rows: the 56 units
  //  of this workgroup are 448 contiguous bytes of a gates row and 224 of a c row)
  // LDS: [h tile][h staging 0][flags][rows][h staging 1][gates staging 0, 1][c staging 0, 1]: the staging tiles are double-buffered by step
  // parity - a step's deferred stores read its tiles while the next step's cell phase fills the other set
  int* rowtab = reinterpret_cast<int*>(smem + CROWS * pitch + CROWS * UW * 2 + 16 + 64);      // [CROWS] row of (sequence, t = 0)
  char* hstage1 = reinterpret_cast<char*>(rowtab + CROWS);
  char* gstage0 = hstage1 + CROWS * UW * 2;                            // [2][CROWS][UW][4 gates] bf16
  char* cstage0 = gstage0 + 2 * CROWS * UW * 8;                        // [2][CROWS][UW] f32
  if (tid < CROWS) {
    int seq = seq0 + tid;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    rowtab[tid] = (int)((seq / p.inner) * p.outer + (seq % p.inner));
  }
  int toff_d = 0;
  bool have_d = false;
  constexpr int SC = UW * 2 / 16;   // 7 chunks of 16 B per staged row
  static_assert(CROWS * SC <= CTHR, "one staged piece per thread");
  constexpr int GC = UW * 8 / 16, CC = UW * 4 / 16;                    // 28 / 14 chunks per gates / c row of this workgroup's units
  static_assert(CROWS * GC == 2 * CTHR && CROWS * CC == CTHR, "two gates pieces and one c piece per thread, every step the same ones");
  // The pieces a thread stores are the same every step, so their byte offsets are computed once (buffer addressing: the step is a scalar
  // offset, a piece that must not be stored - row beyond the cluster's, unit beyond H - lies outside the buffer and is dropped):
  // a deferred store is an LDS read and a buffer store, no arithmetic under the MFMAs.
  constexpr unsigned COOB = 0xFFFFF000u;
  const __amdgpu_buffer_rsrc_t rs_gs = __builtin_amdgcn_make_buffer_rsrc(p.gx, 0, (int)p.g_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs = __builtin_amdgcn_make_buffer_rsrc(p.hout, 0, (int)p.h_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hs2 = __builtin_amdgcn_make_buffer_rsrc(H2 ? p.hout2 : p.hout, 0, (int)p.h_bytes, 0x00020000);
  const int nvu = (H - j * UW) < UW ? (H - j * UW > 0 ? H - j * UW : 0) : UW;     // valid units of this workgroup (a multiple of 8)
  __syncthreads();                                                     // rowtab
  unsigned dvo_g[2], dvo_c, dvo_h;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * CTHR, row = idx / GC, cc = idx - row * GC;
    dvo_g[i] = (row < nrows && cc * 2 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldg_i + (unsigned)(gcol_i + (j * UW + cc * 2) * 4)) * 2u : COOB;
  }
  {
    const int row = tid / CC, cc = tid - row * CC;
    dvo_c = (row < nrows && cc * 4 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldc_i + (unsigned)(hcol_i + j * UW + cc * 4)) * 4u : COOB;
    const int ucol = j * UW + st_cc * 8;
    dvo_h = (tid < CROWS * SC && st_row < nrows && ucol < H) ? ((unsigned)st_grow * (unsigned)ldh_i + (unsigned)(hcol_i + ucol)) * 2u : COOB;
  }
  // The gate pre-activations of the NEXT step travel the same way in the other direction: each thread fetches the two 16-byte pieces it
  // will later store (same offsets), one step ahead, and drops them into the gates staging tile of that step's parity, where a lane picks
  // up its (sequence, unit) 8 bytes before it overwrites them with the activations.  (Fetched per lane - 8 bytes of 16 different rows per
  // wave-instruction in this kernel's accumulator layout - the same 28 KB were 896 quarter-line requests per step: 1.8 us of 7.2.)
  if constexpr (HELP > 0) {
    if (w >= CW) {
      // ---- helper waves: their own time loop, the working waves' two barriers per step ----
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      constexpr int HT = 64 * HELP, NG = CROWS * GC / HT, NC = CROWS * CC / HT;
      static_assert(NG * HT == CROWS * GC && NC * HT == CROWS * CC, "the helper lanes divide the staged pieces evenly");
      const int ht = tid - CTHR;
      unsigned og[NG], oc[NC];
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int idx = ht + i * HT, row = idx / GC, cc = idx - row * GC;
        og[i] = (row < nrows && cc * 2 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldg_i + (unsigned)(gcol_i + (j * UW + cc * 2) * 4)) * 2u : COOB;
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int idx = ht + i * HT, row = idx / CC, cc = idx - row * CC;
        oc[i] = (row < nrows && cc * 4 < nvu) ? ((unsigned)rowtab[row] * (unsigned)ldc_i + (unsigned)(hcol_i + j * UW + cc * 4)) * 4u : COOB;
      }
      // The next step's pre-activations go global -> LDS without passing through registers (buffer_load ... lds: 64 lanes x 16 B land lane-linear
      // at a wave-uniform LDS address - exactly the staging tile's piece order; a piece outside the buffer arrives as zeros).
      const unsigned long gb = (unsigned long)p.gx;
      typedef int rsrc4 __attribute__((ext_vector_type(4)));
      const rsrc4 rg = rsrc4{(int)(unsigned)gb, (int)(unsigned)((gb >> 32) & 0xffffu), (int)p.g_bytes, 0x00020000};
      const unsigned lds_g0 = (unsigned)(size_t)gstage0 + (unsigned)((w - CW) * 64 * 16);
      auto fetch = [&](int par, int toff_) {
        const unsigned soff = (unsigned)(toff_ * ldg_i * 2);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
          const unsigned dst = __builtin_amdgcn_readfirstlane(lds_g0 + (unsigned)(par * (CROWS * UW * 8) + i * HT * 16));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(og[i]), "s"(rg), "s"(soff), "s"(dst) : "memory");
        }
      };
      // The saved gates and c_t of a step: every staged piece into its OWN registers, then the stores; the registers are not written again before
      // the vmcnt(0) at the top of the next step.  (With LDS reads and stores interleaved the compiler recycled a store's four data registers
      // for an LDS read three instructions later and 0.2 % of the stored pieces came out with a zeroed first dword - correct as soon as no
      // register was reused, scripts/diag/dbg_cluster_helpers.py.  h_t -> hout stays with the working waves, which read that piece for the
      // publication anyway.)
      uint4 vg[NG], vc[NC];
      fetch(0, (dir ? p.seq_len - 1 : 0) * stride_i);
      int toff_prev = 0;
      for (int step = 0; step < p.seq_len; ++step) {
        const int t = dir ? (p.seq_len - 1 - step) : step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this step's pre-activations have landed; the previous stores are done
        __builtin_amdgcn_s_barrier();
        const int pp = (step + 1) & 1;                                   // parity of the previous step = of the next one
        const bool st = p.save && step > 0;
#ifndef CABL_NO_DEFERRED
        if (st) {
          const char* gst = gstage0 + pp * (CROWS * UW * 8);
          const char* cst_ = cstage0 + pp * (CROWS * UW * 4);
#pragma unroll
          for (int i = 0; i < NG; ++i) vg[i] = *reinterpret_cast<const uint4*>(gst + (ht + i * HT) * 16);
#pragma unroll
          for (int i = 0; i < NC; ++i) vc[i] = *reinterpret_cast<const uint4*>(cst_ + (ht + i * HT) * 16);
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the tile is in registers: the next step's pre-activations may overwrite it
#ifndef CABL_NO_GX
        if (step + 1 < p.seq_len) fetch(pp, (dir ? t - 1 : t + 1) * stride_i);
#endif
#ifndef CABL_NO_DEFERRED
        if (st) {
#pragma unroll
          for (int i = 0; i < NG; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{vg[i].x, vg[i].y, vg[i].z, vg[i].w}, rs_gs, (int)og[i], toff_prev * ldg_i * 2, 0);
#pragma unroll
          for (int i = 0; i < NC; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{vc[i].x, vc[i].y, vc[i].z, vc[i].w}, rs_cs, (int)oc[i], toff_prev * ldc_i * 4, 0);
        }
#endif
        __builtin_amdgcn_s_barrier();
        toff_prev = t * stride_i;
      }
#ifndef CABL_NO_DEFERRED
      if (p.save) {                                                      // the last step's tiles (complete behind its second barrier)
        const int pp = (p.seq_len + 1) & 1;
        const char* gst = gstage0 + pp * (CROWS * UW * 8);
        const char* cst_ = cstage0 + pp * (CROWS * UW * 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NG; ++i) vg[i] = *reinterpret_cast<const uint4*>(gst + (ht + i * HT) * 16);
#pragma unroll
        for (int i = 0; i < NC; ++i) vc[i] = *reinterpret_cast<const uint4*>(cst_ + (ht + i * HT) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NG; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{vg[i].x, vg[i].y, vg[i].z, vg[i].w}, rs_gs, (int)og[i], toff_prev * ldg_i * 2, 0);
#pragma unroll
        for (int i = 0; i < NC; ++i)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{vc[i].x, vc[i].y, vc[i].z, vc[i].w}, rs_cs, (int)oc[i], toff_prev * ldc_i * 4, 0);
      }
#endif
      return;
    }
  }
  typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
  u32x4g gxl[2];
  auto fetch_gx = [&](int toff_) {
#pragma unroll
    for (int i = 0; i < 2; ++i) gxl[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_gs, (int)dvo_g[i], toff_ * ldg_i * 2, 0);
  };
  if constexpr (HELP == 0) fetch_gx((dir ? p.seq_len - 1 : 0) * stride_i);
  auto deferred_stores = [&](int par) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const char* hst = par ? hstage1 : reinterpret_cast<const char*>(hstage);
    const char* gstage = gstage0 + par * (CROWS * UW * 8);
    const char* cstage = cstage0 + par * (CROWS * UW * 4);
    {
      const uint4 v = *reinterpret_cast<const uint4*>(hst + (tid < CROWS * SC ? tid : 0) * 16);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_hs, (int)dvo_h, toff_d * ldh_i * 2, 0);
      if constexpr (H2) {
        float a0, a1, a2, a3, a4, a5, a6, a7;
        unpack2<f16_t>(v.x, a0, a1); unpack2<f16_t>(v.y, a2, a3); unpack2<f16_t>(v.z, a4, a5); unpack2<f16_t>(v.w, a6, a7);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack2<bf16_t>(a0, a1), pack2<bf16_t>(a2, a3), pack2<bf16_t>(a4, a5), pack2<bf16_t>(a6, a7)},
                                               rs_hs2, (int)dvo_h, toff_d * ldh_i * 2, 0);
      }
    }
    if (p.save && HELP == 0) {                                           // (helper form: the helper waves store the gates and c)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(gstage + (tid + i * CTHR) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_gs, (int)dvo_g[i], toff_d * ldg_i * 2, 0);
      }
      const uint4 v = *reinterpret_cast<const uint4*>(cstage + tid * 16);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_cs, (int)dvo_c, toff_d * ldc_i * 4, 0);
    }
  };
  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? (p.seq_len - 1 - step) : step;
    const int toff = t * stride_i;
    const unsigned pprev = (unsigned)((step + 1) & 1), pcur = (unsigned)(step & 1);
    const unsigned tag_cur = (((unsigned)step >> 1) & 1u) ^ 1u;
    const unsigned tag_prev = (((unsigned)(step - 1) >> 1) & 1u) ^ 1u;
    constexpr int ch = 0;
    const int r0 = 0;
    // 1. h_{t-1} rows -> LDS, each chunk polled until its tags are current
    {
      uint4 hn[HL];
      unsigned pend = 0u;
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        hn[i] = make_uint4(0, 0, 0, 0);
        const int idx = tid + i * CTHR;
        const int row = idx / cpr, cc = idx - row * cpr;
#ifndef CABL_NO_GATHER      // timing diagnostics (wrong results): CABL_NO_GATHER, CABL_NO_MFMA, CABL_NO_CELL, CABL_NO_DEFERRED, CABL_NO_GX, CABL_NO_XSTORE
        if (step > 0 && idx < CROWS * cpr && row < nrows && cc < hchunks) pend |= 1u << i;
#endif
      }
      if (pend && *reinterpret_cast<volatile unsigned*>(deadflag)) pend = 0u;
      const unsigned want = tag_prev ? TAGM : 0u;
      unsigned spins = 0;
      while (pend) {
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          if (pend & (1u << i)) {
            const int idx = tid + i * CTHR;
            const int row = idx / cpr, cc = idx - row * cpr;
            hn[i] = load_sc1(rs, pprev * plane_bytes + cl_bytes + (unsigned)(row * Hp * 2 + cc * 16));
          }
        }
#pragma unroll
        for (int i = 0; i < HL; ++i) {
          if (pend & (1u << i)) {
            const uint4 v = hn[i];
            if ((v.x & TAGM) == want && (v.y & TAGM) == want && (v.z & TAGM) == want && (v.w & TAGM) == want)
              pend &= ~(1u << i);
          }
        }
        if (pend) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 20)) { atomicExch(p.err, 1u); *reinterpret_cast<volatile unsigned*>(deadflag) = 1u; pend = 0u; }
        }
      }
#pragma unroll
      for (int i = 0; i < HL; ++i) {
        const int idx = tid + i * CTHR;
        const int row = idx / cpr, cc = idx - row * cpr;
        uint4 v = hn[i];
        v.x &= ~TAGM; v.y &= ~TAGM; v.z &= ~TAGM; v.w &= ~TAGM;
        if (idx < CROWS * cpr) *reinterpret_cast<uint4*>(htile + row * pitch + cc * 16) = v;
      }
    }
#ifndef CABL_NO_GX
    if constexpr (HELP == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)       // this step's pre-activations -> the staging tile of its parity (this thread read these pieces a step ago)
        *reinterpret_cast<uint4*>(gstage0 + (step & 1) * (CROWS * UW * 8) + (tid + i * CTHR) * 16) = make_uint4(gxl[i][0], gxl[i][1], gxl[i][2], gxl[i][3]);
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (raw barriers: the deferred stores stay in flight across them)
    __builtin_amdgcn_s_barrier();
#ifndef CABL_NO_DEFERRED
    if (have_d) deferred_stores((step + 1) & 1);                         // the previous step's rows: behind this step's gather, under its MFMAs
#endif
    // prefetch the gate pre-activations of the next step (independent of the recurrence)
#ifndef CABL_NO_GX
    if constexpr (HELP == 0) {
      if (step + 1 < p.seq_len) fetch_gx((dir ? t - 1 : t + 1) * stride_i);
    }
#endif
    // 2. gates for (64 rows) x (this wave's quad)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const char* ar = htile + (rt * 16 + lc) * pitch + 16 * lr;
#pragma unroll
      for (int ks = 0; ks < NSLAB; ++ks) {
#ifndef CABL_NO_MFMA
        const uint4 a = *reinterpret_cast<const uint4*>(ar + ks * 64);
        acc = mfma16<TI>(breg[ks], a, acc);
#else
        acc[ks & 3] += __uint_as_float(breg[ks].x);
#endif
      }
      // acc[g] = gate g of unit ul, sequence rt*16 + rl
      const float pre[4] = {acc[0], acc[1], acc[2], acc[3]};
      const uint2 gxv = *reinterpret_cast<const uint2*>(gstage0 + (step & 1) * (CROWS * UW * 8) + (rt * 16 + rl) * (UW * 8) + (w * 4 + ul) * 8);
      float x0, x1, x2, x3;
      unpack2<TI>(gxv.x, x0, x1);
      unpack2<TI>(gxv.y, x2, x3);
      const float gi = pre[0] + x0, gf = pre[1] + x1, gg = pre[2] + x2, go = pre[3] + x3;
#ifdef CABL_NO_CELL
      const float iv = gi, fv = gf, gv = gg, ov = go;
      const float cv = fv * cst[ch][rt] + iv * gv;
      cst[ch][rt] = cv;
      const float hv = uvalid ? ov * cv : 0.f;
#else
      const float iv = sigmoidf_(gi), fv = sigmoidf_(gf), gv = tanhf_(gg), ov = sigmoidf_(go);
      const float cv = fv * cst[ch][rt] + iv * gv;
      cst[ch][rt] = cv;
      const float hv = uvalid ? ov * tanhf_(cv) : 0.f;
#endif
      if (qvalid) reinterpret_cast<TI*>((step & 1) ? hstage1 : reinterpret_cast<char*>(hstage))[(rt * 16 + rl) * UW + w * 4 + ul] = from_f32<TI>(hv);
      if (p.save && qvalid) {
        uint2 gs;
        gs.x = (unsigned)f32_to_bf16(iv) | ((unsigned)f32_to_bf16(fv) << 16);
        gs.y = (unsigned)f32_to_bf16(gv) | ((unsigned)f32_to_bf16(ov) << 16);
        *reinterpret_cast<uint2*>(gstage0 + (step & 1) * (CROWS * UW * 8) + (rt * 16 + rl) * (UW * 8) + (w * 4 + ul) * 8) = gs;
        *reinterpret_cast<float*>(cstage0 + (step & 1) * (CROWS * UW * 4) + (rt * 16 + rl) * (UW * 4) + (w * 4 + ul) * 4) = cv;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // 3. h_t of this workgroup's units -> exchange buffer (tagged); the plain stores follow in the next step (see above)
    const unsigned tagv = tag_cur ? TAGM : 0u;
    if (tid < CROWS * SC) {
      const int row = st_row, cc = st_cc;
      const int ucol = j * UW + cc * 8;
      if (r0 + row < nrows && ucol < H) {
        uint4 v = *reinterpret_cast<const uint4*>(((step & 1) ? hstage1 : reinterpret_cast<const char*>(hstage)) + row * (UW * 2) + cc * 16);
        const uint4 vt = make_uint4(v.x | tagv, v.y | tagv, v.z | tagv, v.w | tagv);
#ifndef CABL_NO_XSTORE
        if (step + 1 < p.seq_len) {
          const unsigned xo = pcur * plane_bytes + cl_bytes + (unsigned)((r0 + row) * Hp * 2 + ucol * 2);
          if (local) store_plain(rs, xo, vt);       // the cluster's consumers read this XCD's L2 (sc1 loads bypass their L1 only)
          else store_sc1(rs, xo, vt);
        }
#endif
      }
    }
    toff_d = toff;
    have_d = true;
  }
  if (have_d) {
    if constexpr (HELP == 0) __syncthreads();                           // (helper form: the last step's second barrier already separates the tiles' writers and readers)
    deferred_stores((p.seq_len + 1) & 1);
  }
}

// quad-ordered recurrent weights: block (dir, quad, slab) = 64 lanes x 16 B; lane (lr, lc): unit quad*4 + (lc>>2),
// gate lc & 3, k = slab*32 + 8*lr + j
template <typename TI>
__device__ __forceinline__ void lstm_pack_quads_dev(const float* __restrict__ whh, TI* __restrict__ out, int H, int Hp) {
  const int nq = (H + 3) >> 2, nslab = Hp / 32, G4 = 4 * H;
  const long total = (long)2 * nq * nslab * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int jj = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % nslab); r /= nslab;
    const int qd = (int)(r % nq);
    const int d = (int)(r / nq);
    const int lc = lane & 15, lr = lane >> 4;
    const int u = qd * 4 + (lc >> 2), g = lc & 3, k = ks * 32 + 8 * lr + jj;
    out[idx] = from_f32<TI>((u < H && k < H) ? whh[((long)d * G4 + g * H + u) * H + k] : 0.f);
  }
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_quads_kernel(const float* __restrict__ whh, TI* __restrict__ out, int H, int Hp) {
  lstm_pack_quads_dev<TI>(whh, out, H, Hp);
}
template <typename TI>
__global__ void __launch_bounds__(256) lstm_pack_quads_multi_kernel(const PackRow* __restrict__ tab, int H, int Hp) {
  const PackRow r = tab[blockIdx.y];
  if (r.whhq) lstm_pack_quads_dev<TI>(r.whh, (TI*)r.whhq, H, Hp);
}

// ---------------------------------------------------------------------------------------------------------------
// Backward through time, cluster form.  Per step (reverse order):
//   A. every lane (row, unit) of the workgroup's 56 units turns dh_out + dh_rec into the four gate gradients, stages them
//      in LDS and publishes its 224 gate columns of the cluster's [64, 4H] dgates tile (sc1 stores); cluster barrier;
//   B. dh_rec[64, 56] = dgates_tile[64, 4H] x W_hh[4H, 56 units]: W_hh^T fragments are register resident, 12 waves =
//      4 unit tiles x 3 K-thirds, the tile is streamed through LDS in two 32-row halves, K-partials are summed via LDS.
// Gate gradients are also written to `gates` (for the weight-gradient GEMMs) after the barrier arrival.
struct ClusterBwdArgs {
  const void* dh; long ldd;
  void* gates; long ldg;
  const float* c;
  const void* whhTq;              // [2][C][4][nslabT][64][16 B]
  bf16_t* dgx;                    // exchange [2 parity][2 dir][ncl][CROWS][4H]
  unsigned* cnt;
  unsigned* err;
  int H;
  long inner, outer, stride;
  int n_seq, seq_len;
  int C, ncl, rows_per_cluster;
};

template <int KPW>
__global__ void __launch_bounds__(CTHR) lstm_bwd_cluster_kernel(ClusterBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane >> 4, lc = lane & 15;
  const int dir = blockIdx.y;
  const int cl = blockIdx.x / p.C, j = blockIdx.x - cl * p.C;
  const int H = p.H, G4 = 4 * H;
  const int pitch = G4 * 2 + 16;
  char* atile = smem;                                               // [32][pitch]  (phase B)  /  stage [64][224] (phase A)
  float* part = reinterpret_cast<float*>(smem + 32 * pitch);        // [3][4][64][16]
  const int nq = (H + 3) >> 2;
  const int qd = j * CW + w;
  const int ul = lc >> 2, q = lc & 3;
  const int lu = w * 4 + ul;                                        // local unit 0..55
  const int u = j * UW + lu;
  const bool uvalid = qd < nq && u < H;
  const int uc = uvalid ? u : H - 1;
  const int nslab = G4 * 2 / 64;
  // phase-B role: unit tile tau, K-third kth
  const int tau = w / 3, kth = w - tau * 3;
  const bool gemm_wave = w < 12;
  const int ks0 = kth * KPW;
  uint4 breg[KPW];
  {
    const char* src = reinterpret_cast<const char*>(p.whhTq) +
                      ((((long)dir * p.C + j) * 4 + (gemm_wave ? tau : 0)) * nslab) * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
      breg[i] = (gemm_wave && ks0 + i < nslab) ? *reinterpret_cast<const uint4*>(src + (long)(ks0 + i) * 1024)
                                               : make_uint4(0, 0, 0, 0);
  }
  const int seq0 = cl * p.rows_per_cluster;
  int seq1 = seq0 + p.rows_per_cluster;
  if (seq1 > p.n_seq) seq1 = p.n_seq;
  const int nrows = seq1 - seq0;
  const bf16_t* dh = reinterpret_cast<const bf16_t*>(p.dh);
  bf16_t* gates = reinterpret_cast<bf16_t*>(p.gates);
  const long gcol0 = (long)dir * G4;
  const unsigned plane_bytes = (unsigned)((long)2 * p.ncl * CROWS * G4 * 2);
  const unsigned cl_bytes = (unsigned)(((long)dir * p.ncl + cl) * CROWS * G4 * 2);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.dgx, 0, (int)(2u * plane_bytes), 0x00020000);
  unsigned* cnt = p.cnt + dir * p.ncl + cl;
  const long prev_off = dir ? p.stride : -p.stride;
  bool dead = false;

  int rowb[4];       // row of (sequence, t = 0); negative = row beyond the cluster (clamped, never stored)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int lrow = rt * 16 + lr * 4 + q;
    int seq = seq0 + lrow;
    if (seq >= p.n_seq) seq = p.n_seq - 1;
    const int rb = (int)((seq / p.inner) * p.outer + (seq % p.inner));
    rowb[rt] = lrow < nrows ? rb : -rb - 1;
  }
  auto rowbase = [&](int rt) -> long { return rowb[rt] >= 0 ? rowb[rt] : -(rowb[rt] + 1); };
  float dcs[4] = {0.f, 0.f, 0.f, 0.f}, dhr[4] = {0.f, 0.f, 0.f, 0.f}, ccur[4];
  uint2 gpre[4];
  float cpre[4], dhpre[4];
  auto prefetch = [&](int step) {     // operands of the pointwise phase of `step` (independent of the recurrence)
    const int t = dir ? step : (p.seq_len - 1 - step);
    const bool first = dir ? (t == p.seq_len - 1) : (t == 0);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const long row = rowbase(rt) + (long)t * p.stride;
      gpre[rt] = *reinterpret_cast<const uint2*>(gates + row * p.ldg + gcol0 + uc * 4);
      const long ci = row * 2 * H + (long)dir * H + uc;
      cpre[rt] = first ? 0.f : p.c[ci + prev_off * 2 * H];
      dhpre[rt] = bf16_to_f32(dh[row * p.ldd + (long)dir * H + uc]);
    }
  };
  {
    const int t0 = dir ? 0 : (p.seq_len - 1);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) ccur[rt] = p.c[(rowbase(rt) + (long)t0 * p.stride) * 2 * H + (long)dir * H + uc];
  }
  prefetch(0);

  for (int step = 0; step < p.seq_len; ++step) {
    const int t = dir ? step : (p.seq_len - 1 - step);
    const long toff = (long)t * p.stride;
    const unsigned plane = (unsigned)(step & 1);
    const bool last = step + 1 == p.seq_len;
    // ---- phase A ----
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const float iv = __uint_as_float(gpre[rt].x << 16), fv = __uint_as_float(gpre[rt].x & 0xffff0000u);
      const float gv = __uint_as_float(gpre[rt].y << 16), ov = __uint_as_float(gpre[rt].y & 0xffff0000u);
      const float dht = dhpre[rt] + dhr[rt];
      const float tc = tanhf_(ccur[rt]);
      const float dct = dcs[rt] + dht * ov * (1.f - tc * tc);
      const float dgi = dct * gv * iv * (1.f - iv), dgf = dct * cpre[rt] * fv * (1.f - fv);
      const float dgg = dct * iv * (1.f - gv * gv), dgo = dht * tc * ov * (1.f - ov);
      dcs[rt] = dct * fv;
      ccur[rt] = cpre[rt];               // c_{t-1} is the next processed step's c_t
      uint2 pk = make_uint2(0u, 0u);
      if (uvalid) {
        pk.x = (unsigned)f32_to_bf16(dgi) | ((unsigned)f32_to_bf16(dgf) << 16);
        pk.y = (unsigned)f32_to_bf16(dgg) | ((unsigned)f32_to_bf16(dgo) << 16);
      }
      *reinterpret_cast<uint2*>(atile + (rt * 16 + lr * 4 + q) * (UW * 8) + lu * 8) = pk;   // stage [64][56 units x 4]
      // gate gradients for the weight-gradient GEMMs
      if (uvalid && rowb[rt] >= 0) *reinterpret_cast<uint2*>(gates + (rowbase(rt) + toff) * p.ldg + gcol0 + u * 4) = pk;
    }
    if (!last) prefetch(step + 1);
    __syncthreads();
    if (!last) {
      constexpr int SC = UW * 8 / 16;   // 28 chunks of 16 B per staged row
      for (int idx = tid; idx < CROWS * SC; idx += CTHR) {
        const int row = idx / SC, cc = idx - row * SC;
        const int colb = j * UW * 8 + cc * 16;
        if (colb >= G4 * 2) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(atile + row * (UW * 8) + cc * 16);
        store_sc1(rs, plane * plane_bytes + cl_bytes + (unsigned)(row * G4 * 2 + colb), v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last) break;
    if (tid == 0) {
      const unsigned target = (unsigned)(step + 1) * (unsigned)p.C;
      unsigned spins = 0;
      while (!dead && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) { dead = true; atomicExch(p.err, 1u); }
      }
    }
    __syncthreads();
    // ---- phase B: dh_rec = dgates_tile x W_hh (this workgroup's units) ----
    const int cpr = G4 * 2 / 16;
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
      f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
      for (int idx = tid; idx < 32 * cpr; idx += CTHR) {
        const int row = idx / cpr, cc = idx - row * cpr;
        const uint4 v = load_sc1(rs, plane * plane_bytes + cl_bytes + (unsigned)((hf * 32 + row) * G4 * 2 + cc * 16));
        *reinterpret_cast<uint4*>(atile + row * pitch + cc * 16) = v;
      }
      __syncthreads();
      if (gemm_wave) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const char* ar = atile + (r2 * 16 + lc) * pitch + 16 * lr;
#pragma unroll
          for (int i = 0; i < KPW; ++i) {
            // slabs past the end have zero B fragments: clamp the A address instead of branching
            const int ks = (ks0 + i < nslab) ? ks0 + i : nslab - 1;
            const uint4 a = *reinterpret_cast<const uint4*>(ar + ks * 64);
            acc[r2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, breg[i]), acc[r2], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            part[((kth * 4 + tau) * CROWS + (hf * 2 + r2) * 16 + lr * 4 + r) * 16 + lc] = acc[r2][r];
      }
      __syncthreads();
    }
    {
      const int tt = lu >> 4, col = lu & 15;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = rt * 16 + lr * 4 + q;
        dhr[rt] = part[((0 * 4 + tt) * CROWS + row) * 16 + col] + part[((1 * 4 + tt) * CROWS + row) * 16 + col] +
                  part[((2 * 4 + tt) * CROWS + row) * 16 + col];
      }
    }
    __syncthreads();
  }
}

// W_hh^T fragments for the cluster BPTT: block (dir, wg j, unit tile tau, slab ks): lane (lr, lc): unit j*56 + tau*16 + lc,
// kk = ks*32 + 8*lr + jj in the gate-interleaved order (u' = kk >> 2, g' = kk & 3)
__global__ void __launch_bounds__(256) lstm_pack_bwd_quads_kernel(const float* __restrict__ whh, bf16_t* __restrict__ out,
                                                                  int H, int C) {
  const int G4 = 4 * H, nslab = G4 / 32;
  const long total = (long)2 * C * 4 * nslab * 64 * 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int jj = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % nslab); r /= nslab;
    const int tau = (int)(r % 4); r /= 4;
    const int j = (int)(r % C);
    const int d = (int)(r / C);
    const int lc = lane & 15, lr = lane >> 4;
    const int lu = tau * 16 + lc, unit = j * UW + lu, kk = ks * 32 + 8 * lr + jj;
    const int up = kk >> 2, gp = kk & 3;
    out[idx] = f32_to_bf16((lu < UW && unit < H && kk < G4) ? whh[((long)d * G4 + gp * H + up) * H + unit] : 0.f);
  }
}

template <int NSLAB, int MAXCH>
static int launch_cluster(const ClusterArgs& p, int f16, hipStream_t st) {
#define URSE_CL_ATTR(...) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_fwd_cluster_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  static bool once = (URSE_CL_ATTR(NSLAB, MAXCH), URSE_CL_ATTR(NSLAB, MAXCH, 2), URSE_CL_ATTR(NSLAB, MAXCH, 2, f16_t, false),
                      URSE_CL_ATTR(NSLAB, MAXCH, 2, f16_t, true), true);
  (void)once;
#undef URSE_CL_ATTR
  const int helpers = getenv("URSE_CLUSTER_HELPERS") ? atoi(getenv("URSE_CLUSTER_HELPERS")) : 2;     // (A/B switch; 0 = the 14-wave form)
  const size_t lds = (size_t)CROWS * lds_frag_pitch(p.Hp * 2) + (size_t)CROWS * UW * 2 + 16 + 64 + CROWS * sizeof(int) + (size_t)CROWS * UW * 14 + (size_t)CROWS * UW * 12;
  dim3 grid(p.C * p.ncl, 2);
  if (f16) {          // (the f16 forward mode exists in the helper-wave form only)
    if (p.hout2) hipLaunchKernelGGL((lstm_fwd_cluster_kernel<NSLAB, MAXCH, 2, f16_t, true>), grid, dim3(CTHR + 128), lds, st, p);
    else hipLaunchKernelGGL((lstm_fwd_cluster_kernel<NSLAB, MAXCH, 2, f16_t, false>), grid, dim3(CTHR + 128), lds, st, p);
  } else if (helpers > 0) hipLaunchKernelGGL((lstm_fwd_cluster_kernel<NSLAB, MAXCH, 2>), grid, dim3(CTHR + 128), lds, st, p);
  else hipLaunchKernelGGL((lstm_fwd_cluster_kernel<NSLAB, MAXCH>), grid, dim3(CTHR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_cluster_fwd");
  return URSE_OK;
}

}  // namespace urse

using namespace urse;

extern "C" int urse_lstm_pack_quads(const float* whh, void* out, int H, int Hp, int dtype, void* stream) {
  URSE_CHECK_ARG(whh && out && H > 0 && Hp % 32 == 0 && Hp >= H && (dtype == URSE_BF16 || dtype == URSE_F16), "urse_lstm_pack_quads: bad argument");
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_quads_kernel<f16_t>, dim3(256), dim3(256), 0, (hipStream_t)stream, whh, (f16_t*)out, H, Hp);
  else hipLaunchKernelGGL(lstm_pack_quads_kernel<bf16_t>, dim3(256), dim3(256), 0, (hipStream_t)stream, whh, (bf16_t*)out, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_quads");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_quads_multi(const void* table, int n_lstm, int H, int Hp, int dtype, void* stream) {
  URSE_CHECK_ARG(table && n_lstm > 0 && n_lstm < 65536 && H > 0 && Hp % 32 == 0 && Hp >= H && (dtype == URSE_BF16 || dtype == URSE_F16),
                 "urse_lstm_pack_quads_multi: bad argument");
  if (dtype == URSE_F16) hipLaunchKernelGGL(lstm_pack_quads_multi_kernel<f16_t>, dim3(256, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, H, Hp);
  else hipLaunchKernelGGL(lstm_pack_quads_multi_kernel<bf16_t>, dim3(256, (unsigned)n_lstm), dim3(256), 0, (hipStream_t)stream, (const PackRow*)table, H, Hp);
  URSE_CHECK_LAUNCH("urse_lstm_pack_quads_multi");
  return URSE_OK;
}

extern "C" int urse_lstm_pack_bwd_quads(const float* whh, void* out, int H, int C, void* stream) {
  URSE_CHECK_ARG(whh && out && H > 0 && H % 8 == 0 && C > 0, "urse_lstm_pack_bwd_quads: bad argument");
  hipLaunchKernelGGL(lstm_pack_bwd_quads_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, whh, (bf16_t*)out, H, C);
  URSE_CHECK_LAUNCH("urse_lstm_pack_bwd_quads");
  return URSE_OK;
}

template <int KPW>
static int launch_cluster_bwd(const ClusterBwdArgs& p, hipStream_t st) {
  static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bwd_cluster_kernel<KPW>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true);
  (void)once;
  size_t lds = (size_t)32 * (4 * p.H * 2 + 16) + (size_t)3 * 4 * CROWS * 16 * 4;
  const size_t stage = (size_t)CROWS * UW * 8;
  if (lds < stage + 3 * 4 * CROWS * 16 * 4) lds = stage + 3 * 4 * CROWS * 16 * 4;
  URSE_CHECK_ARG(lds <= 160 * 1024, "urse_lstm_cluster_bwd: H %d exceeds LDS", p.H);
  dim3 grid(p.C * p.ncl, 2);
  hipLaunchKernelGGL((lstm_bwd_cluster_kernel<KPW>), grid, dim3(CTHR), lds, st, p);
  URSE_CHECK_LAUNCH("urse_lstm_cluster_bwd");
  return URSE_OK;
}

extern "C" int urse_lstm_cluster_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c,
                                     const void* whhTq, void* dgx, void* counters, void* err_flag, int H, int Hp,
                                     int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int reserved_cus,
                                     void* stream) {
  URSE_CHECK_ARG(dh && gates && c && whhTq && dgx && counters && err_flag, "urse_lstm_cluster_bwd: null pointer");
  int64_t plan[6];
  int rc = urse_lstm_cluster_plan(H, Hp, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 4 == 0 && ldd >= 2L * H && ((uintptr_t)dgx % 16) == 0,
                 "urse_lstm_cluster_bwd: bad leading dimension / alignment");
  ClusterBwdArgs p;
  p.dh = dh; p.ldd = ldd; p.gates = gates; p.ldg = ldg; p.c = c; p.whhTq = whhTq; p.dgx = (bf16_t*)dgx;
  p.cnt = (unsigned*)counters; p.err = (unsigned*)err_flag; p.H = H;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  p.C = (int)plan[0]; p.ncl = (int)plan[1]; p.rows_per_cluster = (int)plan[2];
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(counters, 0, sizeof(unsigned) * plan[5], st);
  const int nslab = 4 * H * 2 / 64;
  const int kpw = (nslab + 2) / 3;
  note_launch(URSE_KV_LSTM_BWD_CLUSTER);
  if (kpw <= 2) return launch_cluster_bwd<2>(p, st);
  if (kpw <= 17) return launch_cluster_bwd<17>(p, st);
  set_error("urse_lstm_cluster_bwd: H=%d not supported", H);
  return URSE_ERR_UNSUPPORTED;
}

// workspace query: {C, ncl, rows_per_cluster, rows_pad, hx_elems, n_counters}; returns < 0 if the shape is unsupported
extern "C" int urse_lstm_cluster_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan) {
  URSE_CHECK_ARG(plan && H > 0 && n_seq > 0 && reserved_cus >= 0, "urse_lstm_cluster_plan: bad argument");
  const int nslab = Hp / 32;
  if (Hp % 32 != 0 || !(nslab == 1 || nslab == 2 || nslab == 13) || H % 8 != 0) {
    set_error("urse_lstm_cluster_plan: unsupported H=%d Hp=%d", H, Hp);
    return URSE_ERR_UNSUPPORTED;
  }
  const int nq = (H + 3) / 4;
  const int C = (nq + CW - 1) / CW;
  // 2 directions * ncl * C workgroups, one per CU with a small margin, on the CUs the caller has not promised to other
  // resident work (reserved_cus: workgroups of launches on other streams that run at the same time): all co-resident, or refused
  int ncl = (device_cu_count() - reserved_cus - 4) / 2 / C;
  if (ncl < 1) {
    set_error("urse_lstm_cluster_plan: %d workgroups per cluster do not fit this device (%d CUs reserved)", C, reserved_cus);
    return URSE_ERR_UNSUPPORTED;
  }
  int rpc = (n_seq + ncl - 1) / ncl;
  if (rpc < 1) rpc = 1;
  const int max_rows = CROWS;              // one 64-row chunk per cluster (long-sequence / few-sequence regime)
  if (rpc > max_rows) {
    set_error("urse_lstm_cluster_plan: %d sequences exceed the capacity of the co-resident clusters (%d CUs reserved)", n_seq, reserved_cus);
    return URSE_ERR_UNSUPPORTED;
  }
  ncl = (n_seq + rpc - 1) / rpc;
  const int rows_pad = (rpc + CROWS - 1) / CROWS * CROWS;
  plan[0] = C; plan[1] = ncl; plan[2] = rpc; plan[3] = rows_pad;
  plan[4] = (int64_t)2 * 2 * ncl * rows_pad * Hp;
  plan[5] = 2 * ncl;
  return URSE_OK;
}

extern "C" int urse_lstm_cluster_fwd(void* gx, int64_t ldg, const void* whhq, void* hout, int64_t ldh, float* c, void* hx,
                                     void* counters, void* err_flag, int H, int Hp, int n_seq, int seq_len,
                                     int64_t inner, int64_t outer, int64_t stride, int save, int reserved_cus, int xcd_aware,
                                     int dtype, void* hout_bf16, void* stream) {
  URSE_CHECK_ARG(gx && whhq && hout && hx && counters && err_flag && (c || !save), "urse_lstm_cluster_fwd: null pointer");
  URSE_CHECK_ARG(dtype == URSE_BF16 || dtype == URSE_F16, "urse_lstm_cluster_fwd: operands are bf16 or f16 (dtype %d)", dtype);
  URSE_CHECK_ARG(!hout_bf16 || (dtype == URSE_F16 && ((uintptr_t)hout_bf16 % 16) == 0), "urse_lstm_cluster_fwd: the bf16 copy of h goes with f16 operands only");
  int64_t plan[6];
  int rc = urse_lstm_cluster_plan(H, Hp, n_seq, reserved_cus, plan);
  if (rc) return rc;
  URSE_CHECK_ARG(ldg >= 8L * H && ldg % 4 == 0 && ldh >= 2L * H && (ldh * 2) % 16 == 0 &&
                     ((uintptr_t)hout % 16) == 0 && ((uintptr_t)hx % 16) == 0,
                 "urse_lstm_cluster_fwd: bad leading dimension / alignment");
  URSE_CHECK_ARG(ldg < (1L << 31) && ldh < (1L << 31) && stride * seq_len + (n_seq / inner + 1) * outer < (1L << 31),
                 "urse_lstm_cluster_fwd: row indices must fit 32 bits");
  ClusterArgs p;
  {
    const long rows = stride * (seq_len - 1) + ((n_seq - 1) / inner) * outer + ((n_seq - 1) % inner) + 1;
    URSE_CHECK_ARG(rows * ldg * 2 < 0xFFFFF000L && rows * 2L * H * 4 < 0xFFFFF000L && rows * ldh * 2 < 0xFFFFF000L && ldg % 8 == 0 &&
                       ((uintptr_t)gx % 16) == 0 && (!c || ((uintptr_t)c % 16) == 0),
                   "urse_lstm_cluster_fwd: matrices of %ld rows exceed 32-bit byte offsets, or gx / c are not 16-byte aligned", rows);
    p.g_bytes = (unsigned)(rows * ldg * 2); p.c_bytes = c ? (unsigned)(rows * 2L * H * 4) : 0u; p.h_bytes = (unsigned)(rows * ldh * 2);
  }
  p.gx = gx; p.ldg = ldg; p.whhq = whhq; p.hout = hout; p.hout2 = hout_bf16; p.ldh = ldh; p.c = c; p.hx = (bf16_t*)hx;
  p.cnt = (unsigned*)counters; p.err = (unsigned*)err_flag; p.H = H; p.Hp = Hp; p.save = save;
  p.inner = inner; p.outer = outer; p.stride = stride; p.n_seq = n_seq; p.seq_len = seq_len;
  p.C = (int)plan[0]; p.ncl = (int)plan[1]; p.rows_per_cluster = (int)plan[2]; p.rows_pad = (int)plan[3];
  hipStream_t st = (hipStream_t)stream;
  // the exchange planes start with every tag bit clear (see the hand-off protocol in the kernel)
  (void)hipMemsetAsync(hx, 0, sizeof(bf16_t) * plan[4], st);
  p.xws = nullptr;
  if (xcd_aware && plan[5] >= 9) {       // clusters formed from workgroups that read the same XCC id (registration counters in `counters`)
    (void)hipMemsetAsync(counters, 0, sizeof(unsigned) * plan[5], st);
    p.xws = (unsigned*)counters;
  }
  const int nslab = Hp / 32;
  note_launch(URSE_KV_LSTM_FWD_CLUSTER);
  const int f16 = dtype == URSE_F16;
  if (nslab == 13) return launch_cluster<13, 1>(p, f16, st);
  if (nslab == 2) return launch_cluster<2, 1>(p, f16, st);
  return launch_cluster<1, 1>(p, f16, st);
}
