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
 *  - dtype codes: URSE_F32 / URSE_BF16 select the operand type of the dense contractions
 *    (accumulation is always f32).  Complex tensors are interleaved (re, im) float pairs.
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

#define URSE_WIN_RECT 0
#define URSE_WIN_HANN 1

int urse_version(void);
const char* urse_last_error(void);

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

#ifdef __cplusplus
}
#endif
#endif /* URSE_H_ */
