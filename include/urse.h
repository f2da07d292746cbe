/* liburse_hip -- C ABI of the MI355X-native URGENT-2026 track-1 hot path.
 *
 * The reference (urgent-challenge/urgent2026_challenge_track1) has no native / FFI layer: its
 * "operator API" is the Python call surface of BSRNN_SE / SEModel / the metric functions, whose
 * arithmetic is delegated to espnet2 / torch / pesq / pystoi / fast_bss_eval.  Each entry point
 * below replaces one of those implicit library kernels; the comment on each cites the reference
 * call site it stands behind.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, a negative URSE_ERR_* otherwise; the message is
 *    available from urse_last_error() (thread local).  No C++ exception crosses the ABI.
 *  - all pointers are DEVICE pointers owned by the caller (row-major, contiguous unless a
 *    leading dimension is passed); nothing is allocated per call; workspaces are caller
 *    provided.  Small immutable per-size tables (twiddles, windows, band tables) are built on
 *    first use and cached for the life of the process.
 *  - `stream` is a hipStream_t passed as void*; calls are asynchronous w.r.t. the host.
 *  - dtype codes: URSE_F32 / URSE_BF16 / URSE_F16 select the operand type of the dense contractions
 *    (accumulation is always f32).  Complex tensors are interleaved (re, im) float pairs.
 *    URSE_F16 (IEEE half, 11 significant bits, same bytes and MFMA rate as bf16) is a FORWARD operand format: the kernels that take it
 *    are the forward ones (urse_gemm_nt*, urse_groupnorm_fwd / _apply, urse_bandsplit_norm_fwd, the LSTM forward recurrences and the
 *    weight packs); gradients and every backward operand stay bf16.  Where a forward result is also a backward operand (normalised
 *    inputs, hidden states, the mask decoder's tanh layer) the f16 kernels take an optional `*_bf16` pointer and write the same
 *    values once more in bf16; saved LSTM gate activations are bf16 in both modes.
 *  - no tuning or mode state lives in the library: everything that shapes a launch is an argument of the call
 *    (e.g. target_workgroups of urse_gemm_tn, reserved_cus of the cooperative recurrences).  The only process-wide
 *    mutable data are the read-only-for-results diagnostics below (urse_launch_count) and the caches of immutable tables.
 *  - collectives are NOT part of this ABI (SURVEY 8b lists an `urse_allreduce_bucket`): the gradient all-reduce of
 *    train_se.py:74-83 (Lightning DDP over NCCL) is RCCL reached through the host framework - torch.distributed, backend
 *    "nccl" = RCCL on ROCm - on the contiguous buckets of the flat gradient buffer this library fills (ddp.GradBucketReducer);
 *    wrapping rccl's ncclAllReduce behind a C symbol here would add a communicator-lifetime API and no kernel.
 *  - workspaces: every scratch buffer is an explicit argument whose size follows from the documented shapes (hx / xbuf /
 *    counters from the urse_lstm_*_plan queries, the PESQ slice from urse_pesq_workspace_bytes); kernels that need none
 *    beyond their outputs have no query.
 */
#ifndef URSE_H_
#define URSE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URSE_OK 0
#define URSE_ERR_INVALID_ARG (-1)
#define URSE_ERR_LAUNCH (-2)
#define URSE_ERR_UNSUPPORTED (-3)
#define URSE_ERR_RUNTIME (-4)

#define URSE_F32 0
#define URSE_BF16 1
#define URSE_F16 2            /* IEEE half operands (forward kernels only: gradients stay bf16) */
#define URSE_BF16_ACT_F16 3   /* urse_gemm_tn / urse_gemm_tn_dual only: A (gradients) bf16, B / B2 (the forward's activations) IEEE half,
                                 converted to bf16 in registers behind the LDS fragment read - the weight gradients of an f16-forward step read
                                 x_n and h where the forward left them, no second bf16 copy is written (shapes: urse_gemm_tn_act_f16_supported) */

#define URSE_WIN_RECT 0
#define URSE_WIN_HANN 1

int urse_version(void);
const char* urse_last_error(void);

/* Dispatch bookkeeping (host side, no device work): every dispatcher that chooses between kernel variants counts
 * the variant it launched.  The parity tests of the benchmarked configuration read the counters to prove that the
 * kernels they compared with the oracle are the ones bench.py times.  urse_launch_count: launches of `variant`
 * since the last reset (-1 for an unknown id). */
#define URSE_KV_NT_BRES 0          /* gemm_nt_bres_kernel: weight-stationary gate projection */
#define URSE_KV_NT_RING 1          /* gemm_nt_dma_kernel<.., 256, 2>: LDS-DMA ring, 256-row tiles */
#define URSE_KV_NT_RING_WIDE 2     /* gemm_nt_dma_kernel<.., 128, 4>: 128 x 448 tiles */
#define URSE_KV_NT_128 3           /* gemm_nt_kernel: 128 x 128 register-staged tiles */
#define URSE_KV_NT_GROUPED_RING 4  /* gemm_nt_dma_grouped_kernel */
#define URSE_KV_NT_GROUPED_128 5
#define URSE_KV_TN_RING 6          /* gemm_tn_dma_kernel, single operand */
#define URSE_KV_TN_RING_T 7        /* gemm_tn_dma_kernel on the transposed problem (fc weight gradient) */
#define URSE_KV_TN_DUAL 8          /* gemm_tn_dma_kernel, dual operand (dW_ih + dW_hh in one pass) */
#define URSE_KV_TN_128 9           /* gemm_tn_kernel */
#define URSE_KV_TN_GROUPED 10
#define URSE_KV_LSTM_FWD_STREAM 11 /* lstm_fwd_kernel */
#define URSE_KV_LSTM_FWD_WIDE 12   /* lstm_fwd_wide_kernel */
#define URSE_KV_LSTM_FWD_CLUSTER 13
#define URSE_KV_LSTM_FWD_CLUSTER2 14
#define URSE_KV_LSTM_BWD_STREAM16 15 /* lstm_bwd_kernel, 16 sequences per workgroup */
#define URSE_KV_LSTM_BWD_STREAM32 16 /* lstm_bwd_kernel, 32 sequences per workgroup of 8 waves */
#define URSE_KV_LSTM_BWD_CLUSTER 17
#define URSE_KV_LSTM_BWD_SPLIT 18
#define URSE_KV_STFT960 19         /* register-FFT forward STFT of the 960-point front end */
#define URSE_KV_STFT_GENERIC 20
#define URSE_KV_ISTFT_GENERIC 21
#define URSE_KV_ISTFT960 22
#define URSE_KV_LSTM_FWD_RW 23     /* lstm_fwd_rw_kernel: 16 sequences per wave, weights shared through an LDS-DMA ring */
#define URSE_KV_TN_ACT_F16 24      /* a weight-gradient launch (ring-T or dual) in its URSE_BF16_ACT_F16 form: counted besides its own slot */
#define URSE_KV_LSTM_FWD_RWX 25    /* lstm_fwd_rwx_kernel: row-wave forward with the input projection fused */
#define URSE_KV_LSTM_BWD_NSPLIT 26 /* lstm_bwd_nsplit_kernel: pairs of workgroups split the output columns of the recurrent product */
#define URSE_KV_LSTM_FWD_CLUSTERX 27 /* lstm_fwd_clusterx_kernel: cluster forward with the input projection fused */
/* (28: unused - the three-member N-split BPTT of round 5 measured slower and left the shipped library: csrc/experiments/lstm_nsplit3.hip) */
#define URSE_KV_COUNT 32
int urse_launch_count(int variant);
int urse_launch_counts_reset(void);

/* ---- framed STFT / iSTFT ------------------------------------------------------------------
 * espnet2 Stft.forward / Stft.inverse as called from baseline_code/models/bsrnn.py:37,40 and
 * baseline_code/flow_model.py:136,145 (torch.stft / torch.istft: center, reflect pad, periodic
 * Hann, onesided, not normalised).  T = L / hop + 1, F = n_fft / 2 + 1.
 */
/* wav f32 [B, L] -> spec c64 [B, T, F].  lens (int32 [B], may be NULL): frames t >= olens are zeroed. */
int urse_stft_fwd(const float* wav, const int32_t* lens, float* spec, int B, int L, int n_fft, int hop,
                  int window, void* stream);
/* spec c64 [B, T, F] -> wav f32 [B, L_out]  (torch.istft(length=L_out)). */
int urse_istft_fwd(const float* spec, float* wav, int B, int T, int n_fft, int hop, int L_out, int window,
                   void* stream);
/* adjoint of urse_istft_fwd: grad_wav f32 [B, L_out] -> grad_spec c64 [B, T, F]
 * (PyTorch convention dL/dRe + i dL/dIm). */
int urse_istft_bwd(const float* grad_wav, float* grad_spec, int B, int T, int n_fft, int hop, int L_out,
                   int window, void* stream);

/* ---- dense contractions on the matrix cores (bf16 or exact-f32 MFMA, f32 accumulate) ---------
 * Stand behind cuBLAS under nn.Linear / nn.Conv1d(kernel 1) / the nn.LSTM input projection of
 * espnet2 BSRNN (in-tree twin: baseline_code/models/bsrnn_flowse.py:66-81,296-307) and their
 * autograd backward.  Leading dimensions are in elements; operands 16-byte aligned; K a multiple
 * of 32 (bf16) / 16 (f32) -- callers keep zero-padded channel dims.
 */
/* C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]) (+ resid[M,N] f32).  act: 0 none, 1 tanh,
 * 2 tanh-backward: C = (A*B^T) * (1 - h^2) with h [M,N] (output dtype) passed in the resid slot.
 * Kernel choice is by shape (ring / LDS-resident weights / 128 x 128); all of them add the k-slabs of an output element in the same order, so the
 * choice never changes a bit of the result.  Environment switches for A/Bs: URSE_NT_BRES_MIN_N (LDS-resident weights from this N on, default 448),
 * URSE_NT_WREG_MIN_N (register-resident weights for K = 224, bf16 out; default 0 = off), URSE_NT_NO_DMA (ring kernels off). */
int urse_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias,
                 const float* resid, int64_t ldr, int64_t M, int64_t N, int64_t K, int in_dtype, int out_dtype,
                 int act, void* stream);
/* urse_gemm_nt with f32 output plus the GroupNorm statistics of that output (what the next layer's normalisation needs:
 * espnet BSRNN `norm_time / norm_freq`, GroupNorm(1, N) over a whole batch element): stats[g] = (sum, sum of squares) of rows
 * [g * rows_per_group, (g + 1) * rows_per_group) x N columns; fused into the producing GEMM's epilogue when its ring kernel
 * takes the shape, a separate pass otherwise.  M % rows_per_group == 0, ldc == N, N % 4 == 0. */
int urse_gemm_nt_gnstats(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                         const float* resid, int64_t ldr, int64_t M, int64_t N, int64_t K, int in_dtype, int act,
                         double* stats, int64_t rows_per_group, void* stream);
/* the dgrad GEMM in front of a GroupNorm backward: C[M,N] (f32, dense) = A[M,K] * B[N,K]^T (bf16 operands) and, from the tile while it is on
 * the chip, the sums the GroupNorm backward needs of C = dy against the normalised tensor x ([M,N] f32; groups of rows_per_group rows; stats as
 * urse_groupnorm_fwd wrote them): sums[g] = (sum dy*gamma, sum dy*gamma*xhat) (f64, overwritten), dgamma[c] += sum dy*xhat, dbeta[c] += sum dy.
 * Replaces the reduce pass of urse_groupnorm_bwd (autograd of torch.nn.GroupNorm(1, N) behind espnet2 BSRNN's norm_time / norm_freq, twin
 * baseline_code/models/bsrnn_flowse.py:291,302); follow with urse_groupnorm_bwd_apply.  part: slots*2*N floats of workspace.
 * URSE_ERR_UNSUPPORTED (nothing launched) unless bf16, M >= 2048, 160 <= N <= 224, N % 4 == 0, K % 32 == 0, rows_per_group >= 256 divides M. */
int urse_gemm_nt_gnbwd(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t M, int64_t N, int64_t K, int in_dtype,
                       const float* x, const double* stats, const float* gamma, double* sums, float* dgamma, float* dbeta, float* part,
                       int slots, int64_t rows_per_group, float eps, void* stream);
/* grouped form: `descs` = device array of `groups` records of 12 int64
 * {A, B, C, bias, resid, lda, ldb, ldc, M, N, K, ldr}; grid.x = max_blocks (largest tile count). */
int urse_gemm_nt_grouped(const void* descs, int groups, int max_blocks, int in_dtype, int out_dtype, int act,
                         void* stream);
/* same, with a host mirror of the records: the library validates every group, sizes the grid itself and runs the
 * groups on the LDS-DMA ring kernel when all of them are bf16 with K % 32 == 0, N >= 160, M >= 1024. */
int urse_gemm_nt_grouped_h(const void* descs, const int64_t* host_descs, int groups, int in_dtype, int out_dtype,
                           int act, void* stream);
/* C[Mo,No] (f32) += sum_r A[r,Mo] * B'[r,No]  and optionally colsum[Mo] += sum_r A[r,:].
 * B'[r] = B[r+shift], zero when period > 0 and ((r / inner) % period) == invalid_step
 * (the h_{t-1} operand of the recurrent weight gradient).  perm_h > 0: the columns of A are in the LSTM
 * kernels' gate-interleaved order (dir, unit, gate) and C rows / colsum are written back in nn.LSTM's
 * (dir, gate, unit) order with H = perm_h.  perm_h < 0: flow grad decoder, A columns (bin, 16 sub-channels) ->
 * weight rows (sub-channel, bin) with sb = -perm_h.
 * target_workgroups: workgroups the large-shape kernels aim for (0 = 256, one per CU); a caller that launches beside
 * another kernel passes the CUs that are free.  A per-call argument: the library keeps no tuning state between calls. */
int urse_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                 int64_t R, int64_t Mo, int64_t No, int64_t shift, int64_t inner, int64_t period,
                 int64_t invalid_step, int64_t perm_h, int dtype, int target_workgroups, void* stream);
/* 1 if the URSE_BF16_ACT_F16 form of urse_gemm_tn (No2 == 0; the wide-and-short fc gradient, needs colsum) / urse_gemm_tn_dual (No2 > 0) serves
 * the shape - `d_model.py:61-89`'s backward in an f16-forward step (nn.Linear / nn.LSTM weight gradients, `bsrnn_flowse.py:296-307`). */
int urse_gemm_tn_act_f16_supported(int64_t R, int64_t Mo, int64_t No, int64_t No2, int with_colsum, int64_t inner, int64_t period);
/* Two weight gradients that share their A operand in one pass over A (the two wgrads of one LSTM direction):
 * C[Mo,No] += A^T B (+ colsum) and C2[Mo,No2] += A^T B2', B2' = B2 shifted / masked as in urse_gemm_tn. */
int urse_gemm_tn_dual(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum,
                      const void* B2, int64_t ldb2, float* C2, int64_t ldc2, int64_t R, int64_t Mo, int64_t No, int64_t No2,
                      int64_t shift, int64_t inner, int64_t period, int64_t invalid_step, int64_t perm_h, int dtype,
                      int target_workgroups, void* stream);
/* `groups` independent urse_gemm_tn problems in one launch (per-band weight gradients).  descs = device int64
 * [groups, 24]: {A, B, C, colsum, lda, ldb, ldc, R, Mo, No, shift, inner (>= 1), period, invalid_step,
 * rows_per_slice (multiple of 32; the row range is cut into ceil(R / rows_per_slice) split-R slices), perm_h, 8 x 0};
 * max_blocks >= max over groups of ceil(Mo/128)*ceil(No/128)*slices.  Operand alignment as urse_gemm_tn. */
int urse_gemm_tn_grouped(const void* descs, int groups, int max_blocks, int dtype, void* stream);

/* ---- GroupNorm(1, C) on the channel-last activation stream -----------------------------------
 * espnet2 choose_norm("GN") / choose_norm1d("GN") == nn.GroupNorm(1, N) (in-tree twin:
 * baseline_code/models/bsrnn_flowse.py:291,302 norm_time / norm_freq; :119-136 decoder norms).
 * x f32 [B, T, Kg, W]; a group = (b, kg) spans T rows of W values; channel = col % N;
 * gamma/beta index = kg * gstride + channel.  y rows are [B*T*Kg*(W/N)][Np] (zero padded), dtype
 * URSE_BF16 | URSE_F32.  stats / sums: f64 [B*Kg*2] scratch (sum, sum of squares).  add (may be NULL): f32 [B, N]
 * added after the affine (the flow model's time embedding, bsrnn_flowse.py:293-294). */
int urse_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* add, void* y, double* stats,
                       int B, int T, int Kg, int W, int N, int Np, int gstride, float eps, int out_dtype, void* y_bf16, void* stream);
/* the two halves of urse_groupnorm_fwd on their own: statistics only (accumulates into `stats`, zeroed by the caller), and the
 * normalisation with statistics that exist already (urse_gemm_nt_gnstats) */
int urse_groupnorm_stats(const float* x, double* stats, int B, int T, int Kg, int W, int N, void* stream);
int urse_groupnorm_apply(const float* x, const float* gamma, const float* beta, const float* add, void* y, const double* stats,
                         int B, int T, int Kg, int W, int N, int Np, int gstride, float eps, int out_dtype, void* y_bf16, void* stream);
/* dx = GN backward(dy) (+ dres); dgamma / dbeta are accumulated (+=).  dx_packed (may be NULL): bf16 copy of dx as
 * rows [B*T*Kg*(W/N)][ldp] with columns N..ldp-1 zeroed (the A operand of the next dgrad GEMM, saves a pack pass). */
int urse_groupnorm_bwd(const float* x, const float* dy, const double* stats, const float* gamma, const float* dres,
                       float* dx, float* dgamma, float* dbeta, double* sums, int B, int T, int Kg, int W, int N,
                       int gstride, float eps, void* dx_packed, int ldp, void* stream);
/* the apply pass of urse_groupnorm_bwd alone, with sums that came out of urse_gemm_nt_gnbwd (N % 4 == 0) */
int urse_groupnorm_bwd_apply(const float* x, const float* dy, const double* stats, const double* sums, const float* gamma,
                             const float* dres, float* dx, int B, int T, int Kg, int W, int N, int gstride, float eps, void* dx_packed,
                             int ldp, void* stream);
/* out[out_rows, out_cols] (pitch ldo) = zero-padded copy of in[rows, cols] (or its transpose), with cast. */
int urse_pack2d(const void* in, int64_t ldi, int in_dtype, void* out, int64_t ldo, int out_dtype, int rows, int cols,
                int out_rows, int out_cols, int transpose, void* stream);

/* `nseg` independent pack2d copies in one launch: segs = device int64 [nseg, 8]
 * {in_off, in_rows, in_cols, in_ld, out_off, out_rows, out_cols, out_ld} (elements); f32 source. */
int urse_pack_segments(const float* in, void* out, const void* segs, int nseg, int blocks_per_seg, int transpose,
                       int out_dtype, void* stream);

/* Packs one nn.LSTM(bidirectional)'s f32 weights (wih [2*4H,N], whh [2*4H,H], bih/bhh [2*4H], forward then
 * reverse) into the kernel layouts: wih_p [8H,Np] / wihT_p [N,8H] / bias [8H] (b_ih+b_hh) with gate rows
 * permuted to (dir, unit, gate); whh_frag (2*ceil16(H)*4*Hp elements) and whhT_frag (2*ceil16(H)*4H). */
int urse_lstm_pack(const float* wih, const float* whh, const float* bih, const float* bhh, void* wih_p, void* wihT_p,
                   float* bias, void* whh_frag, void* whhT_frag, int N, int Np, int H, int Hp, int dtype,
                   void* stream);

/* ---- bidirectional LSTM recurrence ---------------------------------------------------------------
 * Sequential part of nn.LSTM(N, 2N, batch_first, bidirectional) (cuDNN under espnet2 BSRNN; twin:
 * baseline_code/models/bsrnn_flowse.py:296-299 rnn_time, :303-306 rnn_freq).  Gate order i,f,g,o.
 * Rows of the [M, .] matrices are addressed as row(s,t) = (s/inner)*outer + s%inner + t*stride.
 *  gx   [M, ldg>=8H]  gate pre-activations x*W_ih^T + b_ih + b_hh, GATE-INTERLEAVED: col = d*4H + unit*4 + gate;
 *                     overwritten with the gate activations when save != 0
 *  whh                recurrent weights in MFMA-fragment order from urse_lstm_pack (K padded to Hp)
 *  hout [M, ldh>=2H]  hidden states, direction d in cols [d*H,(d+1)*H)
 *  c    [M, 2H] f32   cell states (written when save != 0)
 *  rows16: sequences per workgroup / 16 (0 = automatic). */
int urse_lstm_bidir_fwd(void* gx, int64_t ldg, const void* whh, void* hout, int64_t ldh, float* c, int H, int Hp,
                        int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save, int dtype,
                        int rows16, void* hout_bf16, void* stream);
/* Persistent cluster variant of urse_lstm_bidir_fwd (bf16): recurrent weights stay in registers, C workgroups share a
 * set of sequences and exchange h_t through `hx` (zeroed by the call).  Hand-off = "tag in data": every published bf16
 * h carries a step-parity bit in its unused exponent MSB and consumers re-load a chunk until its tags are current (no
 * counters, placement-independent, bounded spins).
 *  whhq: quad-ordered fragments from urse_lstm_pack_quads; plan (urse_lstm_cluster_plan) = {C, clusters per direction,
 *  rows per cluster, padded rows, hx elements (bf16), counters (uint32, used by the BPTT variant only)};
 *  err_flag: uint32 set to 1 if a hand-off timed out (results are then invalid). */
int urse_lstm_pack_quads(const float* whh, void* out, int H, int Hp, int dtype, void* stream);
/* reserved_cus (all cooperative kernels: cluster, cluster2, split): CUs the caller has promised to OTHER work that is resident at
 *  the same time (workgroups of launches on other streams).  The kernels' workgroups wait for each other, so every one of
 *  them must be resident: the plan sizes the grid to device CUs - reserved_cus - margin and the call is REFUSED
 *  (URSE_ERR_UNSUPPORTED, urse_last_error says why) when the sequences do not fit that many co-resident workgroups - the
 *  caller then takes the streaming kernel (urse_lstm_bidir_fwd / _bwd), which has no such requirement. */
int urse_lstm_cluster_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan);
/* xcd_aware != 0: the clusters are formed at kernel start from workgroups that READ the same XCC id from the hardware (each
 *  registers in `counters`, the grid waits for all registrations, every workgroup derives the same assignment): such a cluster
 *  publishes h with stores that stay in its XCD's L2; the workgroups left over form mixed clusters (write-through stores, fewer
 *  sequences).  Placement decides speed only; a pattern of counts the scheme cannot serve uses the static clusters. */
int urse_lstm_cluster_fwd(void* gx, int64_t ldg, const void* whhq, void* hout, int64_t ldh, float* c, void* hx,
                          void* counters, void* err_flag, int H, int Hp, int n_seq, int seq_len, int64_t inner,
                          int64_t outer, int64_t stride, int save, int reserved_cus, int xcd_aware, int dtype, void* hout_bf16,
                          void* stream);
/* Cluster forward with the INPUT PROJECTION FUSED (round 5; the time path of BSRNN at C2: espnet2 BSRNN's rnn_time, reference twin
 * baseline_code/models/bsrnn_flowse.py:296-299): nn.LSTM(bidirectional)'s x W_ih^T + b_ih + h W_hh^T + b_hh in ONE call - no gate-projection GEMM,
 * no [M, 8H] pre-activation matrix written and read back.  xn [M, ldx >= Np] 16-bit normalised input rows (zero K padding); wihq from
 * urse_lstm_pack_quads_x (2 * ceil(H/4) * (Np/32) * 512 elements); bias [2*4H] f32 from urse_lstm_pack; whhq from urse_lstm_pack_quads; gates
 * [M, ldg >= 8H] RECEIVES the bf16 gate activations (save != 0; NULL otherwise); plan / hx / counters / err_flag / reserved_cus / xcd_aware / dtype /
 * hout_bf16 as urse_lstm_cluster_fwd.  urse_lstm_clusterx_supported: Np == 224, Hp == 416 (N = 196, H = 392). */
int urse_lstm_clusterx_supported(int N, int Np, int H, int Hp);
/* Geometry / workspace of urse_lstm_clusterx_fwd: plan[7] = {C, clusters per direction, sequences per cluster, rows_pad, hx elements (16-bit; the exchange
 * planes keep rows at the LDS tile's pitch), counters, ROUNDS}.  Up to (clusters per direction) * 64 sequences per direction the clusters are
 * urse_lstm_cluster_plan's and rounds = 1 (the time path).  Round 6: with more sequences (the band path, rnn_band of the same reference lines: 12,832
 * per direction x 34 steps at C2) every co-resident cluster keeps its weights and takes 64 sequences per ROUND, rounds = ceil(n_seq / (clusters * 64));
 * urse_lstm_clusterx_fwd accepts any n_seq for which this call succeeds.  urse_lstm_clusterx_hx_elems = plan[4]. */
int urse_lstm_clusterx_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan);
int urse_lstm_clusterx_hx_elems(int H, int Hp, int n_seq, int reserved_cus, int64_t* elems);
int urse_lstm_pack_quads_x(const float* wih, void* out, int N, int Np, int H, int dtype, void* stream);
int urse_lstm_clusterx_fwd(const void* xn, int64_t ldx, const void* wihq, const float* bias, const void* whhq, void* gates, int64_t ldg, void* hout,
                           int64_t ldh, float* c, void* hx, void* counters, void* err_flag, int N, int Np, int H, int Hp, int n_seq, int seq_len,
                           int64_t inner, int64_t outer, int64_t stride, int save, int reserved_cus, int xcd_aware, int dtype, void* hout_bf16,
                           void* stream);
/* Generalised cluster forward (csrc/lstm_cluster2.hip): same protocol and arguments, geometry chosen per hidden size
 * (H = 768, the flow model: 24 workgroups per cluster; H = 392: 7).  plan = {C, clusters per direction, rows per cluster,
 * hx bf16 elements}; hx is zeroed by the call; whhq from urse_lstm_pack_quads. */
int urse_lstm_cluster2_plan(int H, int Hp, int n_seq, int reserved_cus, int64_t* plan);
int urse_lstm_cluster2_fwd(void* gx, int64_t ldg, const void* whhq, void* hout, int64_t ldh, float* c, void* hx, void* err_flag,
                           int H, int Hp, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save,
                           int reserved_cus, int dtype, void* stream);      /* dtype: URSE_BF16 | URSE_F16 (gx, whhq, hout; saved gates bf16) */
/* Cluster BPTT (bf16), same protocol: whhTq from urse_lstm_pack_bwd_quads(whh, out [2*C*4*(H/8)*512 bf16], H, C);
 * dgx = exchange buffer of 2*2*ncl*64*4H bf16 elements. */
int urse_lstm_pack_bwd_quads(const float* whh, void* out, int H, int C, void* stream);
int urse_lstm_cluster_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhTq,
                          void* dgx, void* counters, void* err_flag, int H, int Hp, int n_seq, int seq_len, int64_t inner,
                          int64_t outer, int64_t stride, int reserved_cus, void* stream);
/* "Wide" streaming variant of urse_lstm_bidir_fwd (bf16): 64 sequences per workgroup so every streamed weight byte
 * feeds four MFMA row tiles (the band path's 12,832 short sequences).  whhb = block-ordered fragments from
 * urse_lstm_pack_blocks (2*ceil(H/16)*(Hp/32)*4*512 bf16).  `c` [M, 2H] f32 is REQUIRED (it carries c_{t-1} between
 * steps, also when save == 0).  urse_lstm_wide_supported(H, Hp) != 0 tells whether the shape has a kernel. */
int urse_lstm_wide_supported(int H, int Hp);
int urse_lstm_pack_blocks(const float* whh, void* out, int H, int Hp, void* stream);
int urse_lstm_wide_fwd(void* gx, int64_t ldg, const void* whhb, void* hout, int64_t ldh, float* c, int H, int Hp,
                       int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save, void* stream);
/* "Row-wave" variant of urse_lstm_bidir_fwd for many short sequences (bf16; the band path of BSRNN, reference twin
 * baseline_code/models/bsrnn_flowse.py:303-306): a wave owns 16 sequences (h_{t-1} register resident), the seven compute waves
 * of a workgroup share one pass over W_hh per step, streamed L2 -> LDS by a loader wave (csrc/lstm_rw.hip).  No hand-off
 * between workgroups and no co-residency requirement.  whhb from urse_lstm_pack_blocks; `c` is REQUIRED as in
 * urse_lstm_wide_fwd; outputs equal urse_lstm_wide_fwd's bit for bit.  target_workgroups: workgroups the launch is dealt
 * over (0 = one per CU); urse_lstm_rw_supported(H, Hp) != 0 tells whether the shape has a kernel (H = 392).
 * paired != 0: the kernel form that owns two ADJACENT units per lane and block pair (16 / 8 / 4-byte accesses of 256 / 128 / 64
 * contiguous bytes per row); whhb must then come from urse_lstm_pack_blocks_rw, which permutes the weight columns accordingly
 * (same size and fragment layout as urse_lstm_pack_blocks). */
int urse_lstm_rw_supported(int H, int Hp);
int urse_lstm_pack_blocks_rw(const float* whh, void* out, int H, int Hp, void* stream);
int urse_lstm_rw_fwd(void* gx, int64_t ldg, const void* whhb, void* hout, int64_t ldh, float* c, int H, int Hp,
                     int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int save,
                     int target_workgroups, int paired, void* stream);
/* Row-wave forward with the INPUT PROJECTION FUSED (csrc/lstm_rwx.hip): replaces the pair {urse_gemm_nt (x W_ih^T + b -> gx),
 * urse_lstm_rw_fwd} for the band path - nn.LSTM's x W_ih^T + b_ih + h W_hh^T + b_hh in one f32 accumulator per gate, no gx matrix
 * written or read.  xn [M, ldx >= Np] bf16 = the (normalised) LSTM input with zero K padding; wx from urse_lstm_pack_blocks_x
 * (2 * ceil(H/16) * (Hp/32 + Np/32) * 4 * 512 bf16 elements: W_hh and W_ih as one fragment stream); bias [2 * 4H] f32 in the (dir, unit,
 * gate) order of urse_lstm_pack; gates [M, ldg >= 8H] (save != 0) receives the gate ACTIVATIONS the BPTT reads; hout, c, the sequence map,
 * target_workgroups as urse_lstm_rw_fwd.  The pre-activation is not rounded to bf16 between the two products, so results differ
 * from the two-kernel form by bf16 rounding of gx (closer to the f32 reference).  urse_lstm_rwx_supported: N = 196, H = 392. */
/* The pack entry points for ALL LSTMs of a model in one launch each (the model re-packs its 12 BLSTMs after every optimizer step:
 * espnet2 BSRNN keeps nn.LSTM's own layout, here the kernels' fragment orders have to follow the f32 master weights).  table = device array of
 * n_lstm rows of 13 pointers {wih, whh, bih, bhh, wih_p, wihT_p, bias, whh_frag, whhT_frag, whhq, whhb, wx, wihq}: the operands of urse_lstm_pack,
 * then the destinations of urse_lstm_pack_quads / _blocks / _blocks_x / _quads_x (NULL = that LSTM does not use the layout).  Same shapes for all rows. */
int urse_lstm_pack_multi(const void* table, int n_lstm, int N, int Np, int H, int Hp, int dtype, void* stream);
int urse_lstm_pack_quads_multi(const void* table, int n_lstm, int H, int Hp, int dtype, void* stream);
int urse_lstm_pack_quads_x_multi(const void* table, int n_lstm, int N, int Np, int H, int dtype, void* stream);
int urse_lstm_pack_blocks_multi(const void* table, int n_lstm, int H, int Hp, void* stream);
int urse_lstm_pack_blocks_x_multi(const void* table, int n_lstm, int N, int Np, int H, int Hp, int dtype, void* stream);
int urse_lstm_rwx_supported(int N, int Np, int H, int Hp);
int urse_lstm_pack_blocks_x(const float* wih, const float* whh, void* out, int N, int Np, int H, int Hp, int dtype, void* stream);
int urse_lstm_rwx_fwd(const void* xn, int64_t ldx, const void* wx, const float* bias, void* gates, int64_t ldg, void* hout,
                      int64_t ldh, float* c, int N, int Np, int H, int Hp, int n_seq, int seq_len, int64_t inner,
                      int64_t outer, int64_t stride, int save, int target_workgroups, int dtype, void* hout_bf16, void* stream);
/* Split BPTT (bf16) for few, long sequences (time path): 2-3 workgroups share 32 sequences and split the reduction of
 * the recurrent product; f32 partial sums are exchanged through `xbuf` (zeroed by the call) with the tag-in-data
 * hand-off (step parity in the mantissa LSB).  Arguments as urse_lstm_bidir_bwd (whhT from urse_lstm_pack).
 * urse_lstm_split_plan -> {nsplit, clusters per direction, xbuf f32 elements, sequences per cluster (32 | 16)}, < 0 if unsupported (too many sequences
 * for all workgroups to be co-resident, H % 8 != 0, ...).  err_flag: uint32 set to 1 if a hand-off timed out. */
int urse_lstm_split_plan(int H, int n_seq, int reserved_cus, int64_t* plan);
int urse_lstm_split_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT, void* xbuf,
                        void* err_flag, int H, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                        int reserved_cus, void* stream);
/* N-split BPTT (bf16) for few, long sequences (csrc/lstm_nsplit.hip): two workgroups share 32 sequences, each owns half of the hidden
 * units - cell gradients, state and ITS columns of W_hh^T (half the weight stream of urse_lstm_bidir_bwd per step) - and copies the
 * partner's half of the step's gate gradients from the `gates` output (write-through stores + a flag per member and step; the
 * hand-off travels while the member multiplies its own half).  Arguments as urse_lstm_bidir_bwd (whhT from urse_lstm_pack); flags =
 * plan[2] uint32 words (zeroed by the call); err_flag: uint32 set to 1 if a hand-off timed out.  urse_lstm_nsplit_plan ->
 * {pairs per direction, workgroups, flag words}, < 0 if unsupported (H != 392, or the workgroups - which wait for each other - would
 * not be co-resident beside reserved_cus). */
int urse_lstm_nsplit_plan(int H, int n_seq, int reserved_cus, int64_t* plan);
int urse_lstm_nsplit_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT, void* flags,
                         void* err_flag, int H, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride,
                         int reserved_cus, void* stream);
/* Backward through time.  dh [M, ldd>=2H] = gradient w.r.t. hout; gates: in = saved activations,
 * out = gradient w.r.t. the gate pre-activations (same interleaved layout); whhT = fragment-ordered
 * transposed recurrent weights from urse_lstm_pack. */
int urse_lstm_bidir_bwd(const void* dh, int64_t ldd, void* gates, int64_t ldg, const float* c, const void* whhT,
                        int H, int n_seq, int seq_len, int64_t inner, int64_t outer, int64_t stride, int dtype,
                        int rows16, void* stream);

/* ---- band split front end / mask-decoder back end ------------------------------------------------
 * espnet2 BandSplit.forward and the tail of BSRNN.forward / MaskDecoder (in-tree twin
 * baseline_code/models/bsrnn_flowse.py:63-86 and :311-315).  `bands` is an int32 [K, 8] table
 * {f0, sb, xoff, xpad, goff, poff, ppad, 0}: first bin, width, column offset / padded width of the band in
 * the xnb operand, offset into the concatenated norm gamma/beta, column offset / padded width in `pre`. */
/* spec c64 [B,T,F] -> xnb [B*T, ldx]: per-band GroupNorm(1, 2*sb) (zero padded band tail), GEMM-ready. */
int urse_bandsplit_norm_fwd(const float* spec, const int32_t* bands, const float* gamma, const float* beta, void* xnb,
                            double* stats, int B, int T, int F, int K, int ldx, float eps, int out_dtype,
                            void* xnb_bf16, void* stream);
/* dgamma / dbeta (+=) of the band norms from dxnb f32 [B*T, ldx]. */
int urse_bandsplit_norm_bwd(const float* spec, const float* dxnb, const int32_t* bands, const double* stats,
                            float* dgamma, float* dbeta, int B, int T, int F, int K, int ldx, float eps,
                            void* stream);
/* out = GLU(pre_m) * x + GLU(pre_r)  (complex); pre_* f32 [rows, ldp]; f2k int32 [F] bin -> band (-1 none). */
int urse_glu_mask_apply_fwd(const float* pre_m, const float* pre_r, const float* x, float* out,
                            const int32_t* bands, const int32_t* f2k, int64_t rows, int F, int ldp, void* stream);
int urse_glu_mask_apply_bwd(const float* pre_m, const float* pre_r, const float* x, const float* dout, void* dpre_m,
                            void* dpre_r, const int32_t* bands, const int32_t* f2k, int64_t rows, int F, int ldp,
                            int out_dtype, void* stream);
/* x <- x / max|x| * peak in place (baseline_code/inference.py:60); scratch = 4 device bytes. */
int urse_peak_normalize(float* x, int64_t n, float peak, void* scratch, void* stream);
/* ---- on-device dynamic mixing (simulation/simulate_data_from_param.py) -------------------------
 * Batched [B, ld] f32 signals with per-utterance lengths (int32 device arrays); samples >= len are written as 0.
 * urse_nonsilence_power: mean power over the samples espnet2 detect_non_silence(x, 0.01, 1024, 512, boxcar) keeps
 *   (:121-122); hop_scratch = f64 [B * ceil(ld/512)], power = f64 [B].
 * urse_mix_noise (mix_noise :95-126): noise_raw [B, ldn] is wrap-padded (front offset) or cropped (start offset) to the
 *   speech length, scaled to snr_db against the speech, noise_out = scaled noise, noisy_out = speech + noise_out;
 *   scratch = f64 [B * ceil(ld/512) + 2B].
 * urse_fir_full: y = scipy.signal.convolve(x, taps, "full")[:, :len] (add_reverberation :220-230); taps [B or 1, ldt]
 *   with ntaps (device int32 [B or 1]); taps_per_utt != 0 selects one filter per utterance.  x != y.
 * urse_filtfilt_fir: scipy.signal.filtfilt(taps, 1.0, x) (high-pass :461, taps from filter_designs :29-56), ntaps_dev =
 *   device int32 [1] = ntaps; scratch = f32 [2 * B * lds], lds >= ld + 7 * ntaps - 1.
 * urse_quantile_clip (clipping :255-276): np.quantile(x, [qmin, qmax]) with linear interpolation, exact (radix
 *   select), then np.clip in place; bounds = f32 [B, 2] receives the two thresholds.
 * urse_zero_segments (packet_loss :333-341): segments = device int32 [nseg, 3] {utterance, start, end}.
 * urse_joint_peak_scale (:576-584): all three signals *= target / max(|noisy|, |speech|, |noise|, 1e-6) per utterance;
 *   peak_scratch = 4 * B device bytes. */
int urse_nonsilence_power(const float* x, const int32_t* lens, int B, int64_t ld, double threshold, double* hop_scratch,
                          double* power, void* stream);
int urse_mix_noise(const float* speech, const float* noise_raw, const int32_t* noise_lens, int64_t ldn, const int32_t* lens,
                   const int32_t* offsets, const float* snr_db, int B, int64_t ld, float* noise_out, float* noisy_out,
                   double* scratch, void* stream);
int urse_fir_full(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps, int64_t ldt,
                  int taps_per_utt, float* y, void* stream);
/* urse_fft_convolve: the same result as urse_fir_full by power-of-two FFTs of M >= max_len + max_ntaps - 1 <= 2^20 points (float32
 *   transforms: agrees with the direct form to ~1e-6 of the output's peak); max_len / max_ntaps bound lens / ntaps (host values),
 *   workspace from urse_fft_convolve_workspace_bytes. */
int urse_fft_convolve_workspace_bytes(int B, int max_len, int max_ntaps, int64_t* bytes);
int urse_fft_convolve(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps, int64_t ldt,
                      int taps_per_utt, float* y, int max_len, int max_ntaps, void* workspace, int64_t workspace_bytes, void* stream);
int urse_filtfilt_fir(const float* x, const int32_t* lens, int B, int64_t ld, const float* taps, const int32_t* ntaps_dev,
                      int ntaps, float* y, float* scratch, int64_t lds, void* stream);
int urse_quantile_clip(float* x, const int32_t* lens, int B, int64_t ld, const float* qmin, const float* qmax, float* bounds,
                       void* stream);
int urse_zero_segments(float* x, int64_t ld, const int32_t* segments, int nseg, void* stream);
/* resampy.resample (librosa res_type "kaiser_best" / "kaiser_fast" of the bandwidth limitation, :233-252): y[t] = sum of x around
 * t * time_increment weighted by the filter table `win` (right wing, num_table samples per zero crossing, already scaled by the
 * ratio when downsampling) linearly interpolated with `delta` = its first differences; scale = min(1, ratio), index_step =
 * int(scale * num_table); all the index arithmetic in float64 as the package does it.  x [P, ldx] f32 -> y [P, ldy] f32. */
int urse_resample_table(const float* x, int64_t ldx, float* y, int64_t ldy, const double* win, const double* delta, int nwin, int P,
                        int n_orig, int n_out, double time_increment, double scale, int num_table, int index_step, void* stream);
int urse_joint_peak_scale(float* speech, float* noisy, float* noise, int B, int64_t ld, float target, void* peak_scratch,
                          void* stream);
/* y = a*x + b*y (f32). */
int urse_axpby(const float* x, float* y, float a, float b, int64_t n, void* stream);
/* y = torch.nan_to_num(x, nan=0) (baseline_code/flow_model.py:156-157): NaN -> 0, +-inf -> +-FLT_MAX; y may alias x. */
int urse_nan_to_num(const float* x, float* y, int64_t n, void* stream);
/* x *= s[0] with the scalar read from device memory (an upstream autograd scale applied without a host sync). */
int urse_scale_by_device_scalar(float* x, const float* s, int64_t n, void* stream);

/* ---- losses and optimizer ---------------------------------------------------------------------------
 * espnet2 MultiResL1SpecLoss(window_sz, eps, normalize_variance=True, time_domain_weight) and SISNRLoss
 * as constructed in baseline_code/d_model.py:24-25 and called at :74,:80; clip_grad_norm_ + AdamW as
 * configured in train_se.py:78 / d_model.py:104-109. */
/* sums f64 [B,5] = {sum t, sum t^2, sum e, sum e^2, sum e*t}. */
int urse_pair_sums(const float* target, const float* estimate, double* sums, int B, int L, void* stream);
/* loss f32 [B].  `windows` is a HOST array of n_windows even boxcar window sizes (hop = w/2).
 * G (f32 [B,L], may be NULL) receives dLoss/d(a*e) for urse_mrl1_loss_bwd; acc f64 [B,2] scratch. */
int urse_mrl1_loss_fwd(const float* target, const float* estimate, float* loss, float* G, double* sums, double* acc,
                       int B, int L, const int32_t* windows, int n_windows, float eps, float td_weight, void* stream);
/* grad_estimate f32 [B,L] = grad_loss[b] * dLoss_b/d estimate;  c1 f64 [B] scratch. */
int urse_mrl1_loss_bwd(const float* target, const float* estimate, const float* G, const double* sums,
                       const float* grad_loss, float* grad_estimate, double* c1, int B, int L, float eps,
                       void* stream);
/* The same with the reference's NaN-loss guard (baseline_code/d_model.py:75-77: a NaN batch loss is replaced by
 * `se_speech.mean() * 0`, i.e. the step runs on zero gradients): when any loss[b] is NaN the whole gradient is zero. */
int urse_mrl1_loss_bwd_guarded(const float* target, const float* estimate, const float* G, const double* sums,
                               const float* grad_loss, const float* loss, float* grad_estimate, double* c1, int B,
                               int L, float eps, void* stream);
/* loss f32 [B] = 10 log10((1-coh)/coh) (= -SI-SDR in dB), zero-mean. */
int urse_sisnr_fwd(const float* ref, const float* inf, float* loss, double* sums, int B, int L, void* stream);
/* out f64 [1] = sum g^2. */
int urse_grad_sumsq(const float* g, double* out, int64_t n, void* stream);
/* clip_grad_norm_(max_norm) (norm^2 read from device memory, grads pre-multiplied by grad_scale) +
 * AdamW step `step` (1-based); grads are zeroed afterwards when zero_grad != 0; a non-finite norm skips
 * the update. */
int urse_clip_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                         const double* normsq, float max_norm, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float grad_scale, int zero_grad, void* stream);
/* The same step with torch.optim.AdamW's treatment of parameters WITHOUT a gradient (bands above fs/2 get .grad None
 * in the reference, baseline_code/train_se.py:82 find_unused_parameters: no decay, no moment update, no step count).
 * slot u8 [n]: parameter slot of every element (0 = always used).  used f32 [n_slot]: > 0 when the slot received a
 * gradient on some rank this step (it lives behind the flat gradients so the all-reduce sums it; zeroed with them).
 * steps int32 [n_slot]: per-slot AdamW step counts, advanced by the call for the slots it updates.  bias_corr f32
 * [n_slot, 2]: scratch.  skip_flag (uint32, may be NULL): non-zero skips the whole update (a cooperative LSTM kernel
 * reported a timed-out hand-off: the gradients are garbage); a non-finite norm does the same. */
int urse_clip_adamw_step_slots(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const double* normsq, float max_norm, float lr, float beta1, float beta2, float eps,
                               float weight_decay, const uint8_t* slot, float* used, int32_t* steps, float* bias_corr,
                               int n_slot, const uint32_t* skip_flag, float grad_scale, int zero_grad, void* stream);
/* used[i] = i < n_used ? 1 : 0 for i < n_slot: the training forward marks slot 0 and the bands its spectrum holds. */
int urse_fill_used_flags(float* used, int n_slot, int n_used, void* stream);

/* ---- batched intrusive metrics ----------------------------------------------------------------------
 * evaluation_metrics/calculate_intrusive_se_metrics.py: estoi_metric (:37-48) -> pystoi.stoi(extended=True),
 * sdr_metric (:90-109) -> fast_bss_eval.bss_eval_sources(compute_permutation=False, clamp_db=50). */
/* Fourier resampling of whole utterances: y = scipy.signal.resample(x, num) for real f32 rows of arbitrary length n - the
 * `res_type="scipy"` branch of the bandwidth-limitation augmentation (simulate_data_from_param.py:233-252 -> librosa.resample ->
 * scipy.signal.resample: rfft, spectrum cut / zero-padded to num // 2 + 1 bins with the Nyquist rule, irfft, x num / n).
 * Arbitrary lengths run as Bluestein transforms on power-of-two four-step FFTs (csrc/fft_any.hip), one workgroup per row.
 *  plan: per transform length L (n and num each need one): urse_fft_resample_plan_elems(L) -> float2 elements of the plan and of a
 *  scratch buffer; urse_fft_resample_plan(plan, scratch, L) fills it (chirp + transformed conjugate chirp) - build once, reuse.
 *  workspace: urse_fft_resample_workspace_bytes(P, n, num).  2 <= n, num <= 2^19. */
int urse_fft_resample_plan_elems(int n, int64_t* elems, int64_t* tmp_elems);
int urse_fft_resample_plan(void* plan, void* tmp, int n, void* stream);
int urse_fft_resample_workspace_bytes(int P, int n, int num, int64_t* bytes);
int urse_fft_resample(const float* x, int64_t ldx, float* y, int64_t ldy, const void* plan_n, const void* plan_num,
                      void* workspace, int64_t workspace_bytes, int P, int n, int num, void* stream);
/* scipy.signal.resample_poly(x, up, down, window=h) (pystoi.utils.resample_oct): x f32 [P,L] -> y f32 [P,Lout];
 * h_padded (f64, device) = zeros(n_pre_pad) ++ up*h as scipy builds it; n_pre_remove as scipy computes it. */
int urse_resample_poly(const float* x, float* y, const double* h_padded, int hlen, int P, int L, int Lout, int up,
                       int down, int n_pre_remove, void* stream);
/* ESTOI of P pairs already at 10 kHz (f32 [P,L]); out f32 [P].  Scratch: ws_x/ws_y f32 [P,L], tob_x/tob_y f32
 * [P,15,max(1,nfr)] with nfr = ceil((L-256)/128), lens int32 [P], tw512 = 512 complex twiddles e^{-2 pi i j/512}. */
int urse_estoi_batch(const float* ref10k, const float* inf10k, float* out, float* ws_x, float* ws_y, float* tob_x,
                     float* tob_y, int32_t* lens, const float* tw512, int P, int L, void* stream);
/* SDR (dB) of P single-source pairs f32 [P,L] with a 512-tap distortion filter; scratch acf/xcorr f64 [P,512],
 * norms f64 [P,2]. */
int urse_sdr_batch(const float* ref, const float* est, float* out, double* acf, double* xcorr, double* norms, int P,
                   int L, float clamp_db, void* stream);

/* ---- BSRNN-Flow (flow-matching generative model) ---------------------------------------------------------
 * baseline_code/flow_model.py + models/bsrnn_flowse.py + models/odes.py + sampling/ (SURVEY rows a11-a14).
 * Complex tensors are interleaved f32; feature maps are channel-last [B, T, F, C]. */
/* STFTEncoder 'exponent' transform |X|^e e^{j angle X} * factor (inverse != 0: STFTDecoder.spec_back). */
int urse_spec_transform(const float* x, float* y, int64_t n_complex, float exponent, float factor, int inverse,
                        void* stream);
/* xt = (1-t) x0 + t y + sigma(t) z and (cvf may be NULL) cvf = (sigma_max - sigma_min) z + (y - x0)
 * (odes.py:74-98, flow_model.py:164-170); t f32 [B], per_b complex elements per utterance. */
int urse_flow_prepare(const float* x0, const float* y, const float* z, const float* t, float* xt, float* cvf, int B,
                      int64_t per_b, float sigma_min, float sigma_max, void* stream);
/* y += a * x (f32): the Euler update x <- x - step * VF (sampling/odesolvers.py:76-81). */
int urse_axpy(const float* x, float* y, float a, int64_t n, void* stream);
/* GaussianFourierProjection (bsrnn_flowse.py:90-99): out [B, 2*half] = [sin(2 pi t W), cos(2 pi t W)]. */
int urse_time_embedding(const float* t, const float* W, float* out, int B, int half, void* stream);
/* GradDecoder tail (bsrnn_flowse.py:114-117): Conv2d(16 -> 4, 5x5, pad 2) over (F, T); U f32 [B,T,F,16],
 * W [4][16][5][5] (oc, ic, k_F, k_T), pre f32 [B,T,F,4].  bwd: dU (=), dW / dbias (+=). */
int urse_conv5x5_fwd(const float* U, const float* W, const float* bias, float* pre, int B, int T, int F, void* stream);
int urse_conv5x5_bwd(const float* U, const float* W, const float* dpre, float* dU, float* dW, float* dbias, int B, int T,
                     int F, void* stream);
/* out[row, f<F] = sign * (GLU(pre_m) * x + GLU(pre_r)) with GLU over the 4 channels (c0 sig(c2), c1 sig(c3));
 * pre maps are Fs >= F bins wide (bsrnn_flowse.py:311-315; sign = -1 gives FlowSEModel.forward :203-209). */
int urse_glu4_apply_fwd(const float* pre_m, const float* pre_r, const float* x, float* out, int64_t rows, int F,
                        int Fs, float sign, void* stream);
int urse_glu4_apply_bwd(const float* pre_m, const float* pre_r, const float* x, const float* dout, float* dpre_m,
                        float* dpre_r, int64_t rows, int F, int Fs, float sign, void* stream);
/* dst [rows, n] (pitch ldd, bf16|f32) = dU * (1 - U^2): tanh backward of the decoder map, cast for the GEMMs. */
int urse_tanh_bwd_pack(const float* dU, const float* U, void* dst, int64_t rows, int n, int64_t ldd, int out_dtype,
                       void* stream);
/* loss[b] (f64) = 0.5 sum |vf - cvf|^2 (flow_model.py:122-132, 'mse'); grad (may be NULL) = (vf - cvf) * grad_scale. */
int urse_flow_loss(const float* vf, const float* cvf, double* loss, float* grad, int B, int64_t per_b, float grad_scale,
                   void* stream);
/* torch_ema update: shadow -= one_minus_decay * (shadow - params). */
int urse_ema_update(float* shadow, const float* params, float one_minus_decay, int64_t n, void* stream);

/* ---- PESQ ----------------------------------------------------------------------------------------------------------------
 * evaluation_metrics/calculate_intrusive_se_metrics.py:52-88 pesq_metric -> pesq.pesq(fs, ref, deg, mode, on_error=RETURN_VALUES):
 * ITU-T P.862 with the P.862.1 ('nb', fs 8000) / P.862.2 ('wb', fs 16000) MOS-LQO mapping.  One workgroup per pair.
 * ref, deg f32 [pairs, L] (row pitch ld), lens int32 [pairs] or NULL (= L).  mos f32 [pairs]: MOS-LQO, NaN where the
 * reference returns NO_UTTERANCES_DETECTED; raw f32 [pairs] (may be NULL): the raw P.862 score.  trace int32 [pairs, 286]:
 * the integer outputs of the alignment stages {crude delay, utterances, first / last frame, bad intervals, three words of stage
 * timers (six 16-bit counts of 64 us: level filters, input filter, DC + alignment IIR, VAD, alignment + splitting, model),
 * utterance start[50], end[50], delay[50], bad-interval (first, last) frame[64]} - what "bit-exact through the integer
 * stage" is checked on.  workspace: urse_pesq_workspace_bytes(pairs, L, fs) bytes of device memory. */
#define URSE_PESQ_TRACE 286
int urse_pesq_workspace_bytes(int pairs, int L, int fs, int64_t* bytes);
int urse_pesq_batch(const float* ref, const float* deg, int64_t ld, const int32_t* lens, int pairs, int L, int fs, int wb,
                    float* mos, float* raw, int32_t* trace, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- FLAC decoding (HOST code, HOST pointers) -------------------------------------------------------------------
 * soundfile.read behind baseline_code/dataset.py:318-322 and simulation/simulate_data_from_param.py:347-349 (libsndfile is
 * not in the image; URGENT sources are largely FLAC).  info int64 [6] = {fs, channels, bits, total frames | 0, min block,
 * max block}; urse_flac_decode writes interleaved int32 samples [capacity_frames, channels] and the number of frames. */
int urse_flac_info(const void* data, int64_t nbytes, int64_t* info);
int urse_flac_decode(const void* data, int64_t nbytes, int32_t* out, int64_t capacity_frames, int64_t* decoded);

/* ---- diagnostics (not on the product path) ----------------------------------------------------------------------
 * streams `bytes` of `buf` with `width`-byte (4 | 8 | 16) per-lane reads (write = 0) or writes (write = 1): a known byte
 * count to calibrate the rocprofv3 FETCH_SIZE / WRITE_SIZE counters per access width (scripts/pmc_calibrate.py). */
int urse_diag_stream(void* buf, float* sink, int64_t bytes, int width, int write, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* URSE_H_ */
