"""Host-side operators of the hot path: thin torch.autograd wrappers over the C ABI.

PyTorch supplies device memory, streams and autograd bookkeeping only; every
arithmetic step runs in a hand-written gfx950 kernel of liburse_hip.so.
"""
import os

import torch

from . import _lib
from ._lib import call, require_cuda, stream_ptr

WIN_RECT, WIN_HANN = 0, 1
F32, BF16, F16 = 0, 1, 2       # include/urse.h: URSE_F32 / URSE_BF16 / URSE_F16
BF16_ACT_F16 = 3               # URSE_BF16_ACT_F16: weight-gradient GEMMs with bf16 gradients against the forward's f16 activations
HALF_TYPES = (torch.bfloat16, torch.float16)   # 16-bit operand formats: same kernels, layouts and padding; f16 is the FORWARD-only format


def bwd_dtype(dtype):
    """operand type of the backward kernels for a forward operand type: gradients need bf16's range, so the f16 forward mode
    (compute_dtype "f16": 11 significant bits, what north_star's 1e-3 on the enhanced waveform needs) keeps bf16 backward operands."""
    return torch.bfloat16 if dtype == torch.float16 else dtype

# Optional per-kernel HIP-event timing (bench.py): name -> [(start_event, end_event), ...]
_timing = None


def enable_timing(names):
    global _timing
    _timing = {n: [] for n in names}


def disable_timing():
    global _timing
    t, _timing = _timing, None
    return t


def timed_call(tname, name, *args):
    if _timing is not None and tname in _timing:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        call(name, *args)
        b.record()
        _timing[tname].append((a, b))
    else:
        call(name, *args)


# kernel variants the dispatchers of the library count (include/urse.h: URSE_KV_*)
KERNEL_VARIANTS = ("nt_bres", "nt_ring", "nt_ring_wide", "nt_128", "nt_grouped_ring", "nt_grouped_128", "tn_ring", "tn_ring_t",
                   "tn_dual", "tn_128", "tn_grouped", "lstm_fwd_stream", "lstm_fwd_wide", "lstm_fwd_cluster",
                   "lstm_fwd_cluster2", "lstm_bwd_stream16", "lstm_bwd_stream32", "lstm_bwd_cluster", "lstm_bwd_split",
                   "stft960", "stft_generic", "istft_generic", "istft960", "lstm_fwd_rw", "tn_act_f16", "lstm_fwd_rwx", "lstm_bwd_nsplit",
                   "lstm_fwd_clusterx", "_unused28")


_PAGEABLE_UPLOADS = os.environ.get("URSE_PAGEABLE_UPLOADS", "0") == "1"


def upload(host_tensor, device, cached=False):
    """small host table -> device through page-locked memory, non-blocking: a pageable `.to(device)` makes the host wait for
    everything queued on the stream (the per-band descriptor tables go up eight times per step).
    cached=True: the table goes into a process-wide cache that OTHER streams will read without an event (filter tables, resampler
    tables): the copy is waited for once, here, so that no consumer can see a half-written table whatever stream touched the cache
    first (the first toucher is often the lowest-priority prefetch stream with two batches of work queued)."""
    if torch.device(device).type != "cuda":
        return host_tensor
    if _PAGEABLE_UPLOADS:          # A/B switch: the blocking copies this replaced
        return host_tensor.to(device)
    out = host_tensor.contiguous().pin_memory().to(device, non_blocking=True)
    if cached:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ev.synchronize()
    return out


def launch_counts(reset=False):
    """{variant: launches since the last reset}: which kernels the dispatchers actually picked (parity tests of the
    benchmarked configuration assert on it)."""
    lib = _lib.load()
    out = {n: lib.urse_launch_count(i) for i, n in enumerate(KERNEL_VARIANTS)}
    if reset:
        lib.urse_launch_counts_reset()
    return out


def _f32c(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


def stft_forward(wav, n_fft, hop, window=WIN_HANN, lens=None):
    """espnet Stft.forward (bsrnn.py:37): wav f32 [B, L] -> complex64 [B, T, F]."""
    require_cuda(wav)
    wav = _f32c(wav)
    B, L = wav.shape
    T, Fb = L // hop + 1, n_fft // 2 + 1
    spec = torch.empty(B, T, Fb, 2, device=wav.device, dtype=torch.float32)
    if lens is not None:
        if lens.device.type == "cpu":   # through page-locked memory: a pageable copy makes the host wait for the whole queue
            lens = upload(lens.to(torch.int32), wav.device)
        else:
            lens = lens.to(device=wav.device, dtype=torch.int32).contiguous()
    timed_call("stft_fwd", "stft_fwd", wav, lens, spec, B, L, n_fft, hop, window, stream_ptr())
    return torch.view_as_complex(spec)


class _ISTFT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec_ri, n_fft, hop, length, window):
        B, T, Fb, _ = spec_ri.shape
        wav = torch.empty(B, length, device=spec_ri.device, dtype=torch.float32)
        call("istft_fwd", spec_ri, wav, B, T, n_fft, hop, length, window, stream_ptr())
        ctx.cfg = (B, T, Fb, n_fft, hop, length, window)
        return wav

    @staticmethod
    def backward(ctx, g):
        B, T, Fb, n_fft, hop, length, window = ctx.cfg
        g = _f32c(g)
        gs = torch.empty(B, T, Fb, 2, device=g.device, dtype=torch.float32)
        call("istft_bwd", g, gs, B, T, n_fft, hop, length, window, stream_ptr())
        return gs, None, None, None, None


def istft_forward(spec, n_fft, hop, length, window=WIN_HANN):
    """espnet Stft.inverse (bsrnn.py:40): complex64 [B, T, F] (or real [B,T,F,2]) -> wav f32 [B, length]."""
    if spec.is_complex():
        spec = torch.view_as_real(spec)
    require_cuda(spec)
    return _ISTFT.apply(_f32c(spec), n_fft, hop, int(length), window)


# ---------------------------------------------------------------------------------------------
# dense contractions
# ---------------------------------------------------------------------------------------------
def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise TypeError("URSE GEMM operands are bf16, f16 or f32, got %s" % t.dtype)


def dtype_code(dtype):
    """URSE_* code of a torch dtype."""
    return {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[dtype]


def gemm_nt(A, W, bias=None, resid=None, act=0, out=None, out_dtype=None, N=None, gn_rows=0):
    """out[M, N] = act(A[M, K] @ W[N, K]^T + bias) (+ resid).  A, W: 2-D, unit inner stride, same dtype.
    f16 operands with act = 1 and a 16-bit output: `resid` (a bf16 tensor shaped like `out`) receives the same values in bf16.
    gn_rows > 0 (f32 dense output): also returns the GroupNorm statistics of `out` per group of gn_rows rows, f64 [M / gn_rows, 2]
    (sum, sum of squares) as `groupnorm_fwd(..., stats=)` takes them -> (out, stats)."""
    require_cuda(A, W)
    M, K = A.shape
    Nw = W.shape[0] if N is None else N
    assert W.shape[1] == K and A.stride(1) == 1 and W.stride(1) == 1 and A.dtype == W.dtype
    if out is None:
        out = torch.empty(M, Nw, device=A.device, dtype=out_dtype or A.dtype)
    assert out.stride(1) == 1
    if gn_rows:
        assert out.dtype == torch.float32 and out.stride(0) == Nw and M % gn_rows == 0
        stats = torch.empty(M // gn_rows * 2, device=A.device, dtype=torch.float64)
        call("gemm_nt_gnstats", A, A.stride(0), W, W.stride(0), out, out.stride(0), bias, resid,
             0 if resid is None else resid.stride(0), M, Nw, K, _dt(A), act, stats, gn_rows, stream_ptr())
        return out, stats
    call("gemm_nt", A, A.stride(0), W, W.stride(0), out, out.stride(0), bias, resid,
         0 if resid is None else resid.stride(0), M, Nw, K, _dt(A), _dt(out), act, stream_ptr())
    return out


def gemm_tn(A, Bm, out, colsum=None, Mo=None, No=None, shift=0, inner=1, period=0, invalid_step=0, perm_h=0, target_wgs=0):
    """out[Mo, No] (f32) += A[R, Mo]^T @ B'[R, No]; optional colsum[Mo] += sum_r A[r].  target_wgs: workgroups to aim for when
    the launch shares the chip with another kernel (0 = one per CU)."""
    require_cuda(A, Bm, out)
    R = A.shape[0]
    Mo = A.shape[1] if Mo is None else Mo
    No = Bm.shape[1] if No is None else No
    assert Bm.shape[0] == R and A.stride(1) == 1 and Bm.stride(1) == 1
    assert out.dtype == torch.float32 and out.stride(-1) == 1
    call("gemm_tn", A, A.stride(0), Bm, Bm.stride(0), out, out.stride(0), colsum, R, Mo, No, shift, inner, period,
         invalid_step, perm_h, _tn_dt(A, Bm), int(target_wgs), stream_ptr())
    return out


def _tn_dt(A, *Bs):
    """operand code of a weight-gradient GEMM: the common dtype, or BF16_ACT_F16 for bf16 gradients against f16 activations (f16-forward training:
    the forward's x_n / h are read where they are and converted to bf16 in the kernel's registers - no second copy is written)."""
    if A.dtype == torch.bfloat16 and all(b.dtype == torch.float16 for b in Bs):
        return BF16_ACT_F16
    assert all(b.dtype == A.dtype for b in Bs), (A.dtype, [b.dtype for b in Bs])
    return _dt(A)


# f16-forward training: weight gradients read the f16 activations directly where the library has the mixed-operand kernel for the shape;
# 0: the producing kernels write x_n and h a second time in bf16 (round 5: +3 to 4 % on the step, 12.5 GB of extra writes)
TN_ACT_F16 = os.environ.get("URSE_TN_ACT_F16", "1") != "0"


def tn_act_f16_supported(R, Mo, No, No2=0, with_colsum=True, inner=1, period=0):
    return TN_ACT_F16 and bool(_lib.load().urse_gemm_tn_act_f16_supported(int(R), int(Mo), int(No), int(No2), int(bool(with_colsum)),
                                                                        int(inner), int(period)))


def gemm_tn_dual(A, Bm, out, colsum, B2, out2, Mo, No, No2, shift, inner, period, invalid_step, perm_h=0, target_wgs=0):
    """out[Mo, No] += A^T Bm (+ colsum) and out2[Mo, No2] += A^T B2' (shifted / masked) in one pass over A."""
    require_cuda(A, Bm, B2, out, out2)
    R = A.shape[0]
    assert Bm.shape[0] == R and B2.shape[0] == R and A.stride(1) == 1 and Bm.stride(1) == 1 and B2.stride(1) == 1
    assert out.dtype == torch.float32 and out2.dtype == torch.float32
    call("gemm_tn_dual", A, A.stride(0), Bm, Bm.stride(0), out, out.stride(0), colsum, B2, B2.stride(0), out2, out2.stride(0), R,
         Mo, No, No2, shift, inner, period, invalid_step, perm_h, _tn_dt(A, Bm, B2), int(target_wgs), stream_ptr())


def tn_desc(A, Bm, out, colsum=None, Mo=None, No=None, shift=0, inner=1, period=0, invalid_step=0, perm_h=0):
    """one row of a urse_gemm_tn_grouped descriptor table (rows_per_slice is filled in by gemm_tn_grouped)."""
    R = A.shape[0]
    Mo = A.shape[1] if Mo is None else Mo
    No = Bm.shape[1] if No is None else No
    assert Bm.shape[0] == R and A.stride(1) == 1 and Bm.stride(1) == 1 and A.dtype == Bm.dtype
    assert out.dtype == torch.float32 and out.stride(-1) == 1 and out.stride(0) >= No
    # the kernel reads whole 16-byte chunks that START below Mo / No (A and Bm may be column slices of wider matrices): the last
    # row must hold round_up(Mo | No, 16 / itemsize) elements inside its storage
    epc = 16 // A.element_size()
    for t, n in ((A, Mo), (Bm, No)):
        last = t.storage_offset() + (R - 1) * t.stride(0) + (n + epc - 1) // epc * epc
        assert last * t.element_size() <= t.untyped_storage().nbytes(), "tn_desc: operand rows are not padded to 16 bytes"
    return [A.data_ptr(), Bm.data_ptr(), out.data_ptr(), 0 if colsum is None else colsum.data_ptr(), A.stride(0),
            Bm.stride(0), out.stride(0), R, Mo, No, shift, max(1, inner), period, invalid_step, 0, perm_h] + [0] * 8


def gemm_tn_grouped(rows, dtype, device, target_blocks=1024):
    """out_g[Mo, No] += A_g^T @ B_g for every descriptor row (see tn_desc) in ONE launch."""
    bkr = 32 if dtype == torch.bfloat16 else 16
    tiles = [((r[8] + 127) // 128) * ((r[9] + 127) // 128) for r in rows]
    per_group = max(1, target_blocks // max(1, sum(tiles)))
    max_blocks = 0
    for r, tl in zip(rows, tiles):
        R = r[7]
        slices = max(1, min(per_group, (R + 4 * bkr - 1) // (4 * bkr)))
        rps = (R + slices - 1) // slices
        rps = (rps + bkr - 1) // bkr * bkr
        r[14] = rps
        max_blocks = max(max_blocks, tl * ((R + rps - 1) // rps))
    descs = upload(torch.tensor(rows, dtype=torch.int64), device)
    call("gemm_tn_grouped", descs, len(rows), max_blocks, BF16 if dtype == torch.bfloat16 else F32, stream_ptr())


# ---------------------------------------------------------------------------------------------
# GroupNorm / packing / LSTM recurrence (raw, non-autograd wrappers; bsrnn.py composes them)
# ---------------------------------------------------------------------------------------------
def pad_to(n, m):
    return (n + m - 1) // m * m


def kpad(n, dtype):
    """zero-padded contraction length the MFMA GEMMs need: multiple of 32 (bf16 / f16) / 16 (f32)."""
    return pad_to(n, 32 if dtype in HALF_TYPES else 16)


def pack2d(inp, out_rows, out_cols, dtype, transpose=False, out=None):
    """zero-padded (optionally transposed) cast copy of a 2-D tensor."""
    require_cuda(inp)
    assert inp.dim() == 2 and inp.stride(1) == 1
    rows, cols = (inp.shape[1], inp.shape[0]) if transpose else inp.shape
    if out is None:
        out = torch.empty(out_rows, out_cols, device=inp.device, dtype=dtype)
    call("pack2d", inp, inp.stride(0), _dt(inp), out, out.stride(0), _dt(out), rows, cols, out_rows, out_cols,
         int(transpose), stream_ptr())
    return out


def groupnorm_fwd(x, gamma, beta, B, T, Kg, W, N, Np, gstride, dtype, eps=1e-5, add=None, stats=None, bf16_copy=False):
    """x f32 [B,T,Kg,W] -> (y [B*T*Kg*(W/N), Np] dtype, stats f64 [B*Kg*2]).  stats: the statistics of x where its producer
    computed them already (`gemm_nt(..., gn_rows=)`): only the normalisation runs.
    bf16_copy (dtype f16, training): -> (y, stats, y_bf16), the rows once more in bf16 for the weight-gradient GEMMs."""
    require_cuda(x)
    y = torch.empty(B * T * Kg * (W // N), Np, device=x.device, dtype=dtype)
    y2 = torch.empty_like(y, dtype=torch.bfloat16) if (bf16_copy and dtype == torch.float16) else None
    if stats is not None:
        assert stats.numel() == B * Kg * 2 and stats.dtype == torch.float64
        call("groupnorm_apply", x, gamma, beta, add, y, stats, B, T, Kg, W, N, Np, gstride, float(eps), _dt(y), y2, stream_ptr())
    else:
        stats = torch.empty(B * Kg * 2, device=x.device, dtype=torch.float64)
        call("groupnorm_fwd", x, gamma, beta, add, y, stats, B, T, Kg, W, N, Np, gstride, float(eps), _dt(y), y2, stream_ptr())
    if bf16_copy is None:            # (callers that always unpack three: the third is the backward's operand, y itself without a copy)
        return y, stats, y
    return (y, stats, (y2 if y2 is not None else y)) if bf16_copy else (y, stats)


def groupnorm_bwd(x, dy, stats, gamma, dres, dgamma, dbeta, B, T, Kg, W, N, gstride, eps=1e-5, pack_ld=0, sums=None):
    """returns dx f32 (same layout as x); dgamma/dbeta accumulated in place.  pack_ld > 0: also returns the bf16 copy of dx
    as rows of pack_ld columns (zero padded) -> (dx, dx_packed).  sums: the reduce pass's results when the GEMM that produced dy
    computed them (gemm_nt_gnbwd; dgamma / dbeta are accumulated there) - only the apply pass runs."""
    dx = torch.empty_like(x)
    dxp = torch.empty(x.numel() // N, pack_ld, device=x.device, dtype=torch.bfloat16) if pack_ld else None
    if sums is not None:
        call("groupnorm_bwd_apply", x, dy, stats, sums, gamma, dres, dx, B, T, Kg, W, N, gstride, float(eps), dxp, pack_ld, stream_ptr())
        return (dx, dxp) if pack_ld else dx
    sums = torch.empty(B * Kg * 2, device=x.device, dtype=torch.float64)
    call("groupnorm_bwd", x, dy, stats, gamma, dres, dx, dgamma, dbeta, sums, B, T, Kg, W, N, gstride, float(eps), dxp,
         pack_ld, stream_ptr())
    return (dx, dxp) if pack_ld else dx


FUSE_GN_BWD = os.environ.get("URSE_FUSE_GN_BWD", "1") != "0"      # GroupNorm-backward sums on the dgrad GEMM's epilogue
GN_BWD_SLOTS = 16


def gemm_nt_gnbwd_supported(M, N, K, rows_per_group, dtype):
    return (dtype == torch.bfloat16 and M >= 2048 and 160 <= N <= 224 and N % 4 == 0 and K % 32 == 0 and K >= 96 and
            rows_per_group >= 256 and M % rows_per_group == 0 and not os.environ.get("URSE_NT_NO_DMA"))


def gemm_nt_gnbwd(A, W, N, x, stats, gamma, dgamma, dbeta, rows_per_group, eps=1e-5):
    """dy[M, N] f32 = A[M, K] @ W[N.., K]^T plus the GroupNorm-backward reduce of dy against x (f32 [M, N], same row order) in the same
    launch -> (dy, sums) with sums for groupnorm_bwd(sums=); dgamma / dbeta are accumulated here.  Check gemm_nt_gnbwd_supported first."""
    require_cuda(A, W, x)
    M, K = A.shape
    assert W.shape[1] == K and A.stride(1) == 1 and W.stride(1) == 1 and A.dtype == W.dtype and x.is_contiguous() and x.numel() == M * N
    dy = torch.empty(M, N, device=A.device, dtype=torch.float32)
    ng = M // rows_per_group * 2
    ws = torch.empty(ng + GN_BWD_SLOTS * N, device=A.device, dtype=torch.float64)      # sums, then the per-channel slots (f32): one fill
    sums, part = ws[:ng], ws[ng:].view(torch.float32)
    call("gemm_nt_gnbwd", A, A.stride(0), W, W.stride(0), dy, M, N, K, _dt(A), x, stats, gamma, sums, dgamma, dbeta, part, GN_BWD_SLOTS,
         rows_per_group, float(eps), stream_ptr())
    return dy, sums


def model_lstm_layouts():
    """the optional weight layouts the model's dispatch can reach with the current switches (the model re-packs 12 LSTMs after every
    optimizer step: a layout nobody reads is a launch and a few MB of writes per LSTM and step)."""
    lay = set()
    if USE_CLUSTER_LSTM:
        lay.add("whhq")
        if USE_CLUSTERX_LSTM:
            lay.add("wihq")
    if USE_CLUSTER_LSTM_BWD:
        lay.add("whhTq")
    if USE_RW_LSTM and USE_RWX_LSTM:
        lay.add("wx")
    if USE_WIDE_LSTM or USE_RW_LSTM:
        lay.add("whhb")          # (the unfused row-wave / wide forward; also what a shape without a fused kernel falls back to)
    if USE_RW_LSTM and RW_PAIRED and not USE_RWX_LSTM:
        lay.add("whhb_rw")
    return lay


def lstm_pack(wih, whh, bih, bhh, N, H, dtype, out=None, layouts=None):
    """f32 nn.LSTM weights (fwd+reverse concatenated) -> dict of kernel-layout operands (see urse_lstm_pack).
    layouts: which of the optional layouts to produce ({"whhq", "whhTq", "whhb", "whhb_rw", "wx", "wihq"}; None = all the shape supports)."""
    want = lambda name: layouts is None or name in layouts
    dev = wih.device
    Np, Hp = kpad(N, dtype), kpad(pad_to(H, 16), dtype)
    nu = pad_to(H, 16)
    bdt = bwd_dtype(dtype)        # dtype f16: the forward layouts (wih, whh, whhq, wx) are f16, the backward ones (wihT, whhT) bf16
    if out is None:
            out = dict(wih=torch.empty(8 * H, Np, device=dev, dtype=dtype),
                   wihT=torch.empty(N, 8 * H, device=dev, dtype=bdt),
                   bias=torch.empty(8 * H, device=dev, dtype=torch.float32),
                   whh=torch.empty(2 * nu * 4 * Hp, device=dev, dtype=dtype),
                   whhT=torch.empty(2 * nu * 4 * H, device=dev, dtype=bdt), Np=Np, Hp=Hp)
    call("lstm_pack", wih, whh, bih, bhh, out["wih"], out["wihT"], out["bias"], out["whh"], out["whhT"], N, Np, H, Hp,
         _dt(out["wih"]), stream_ptr())
    if dtype == torch.float16 and Hp % 32 == 0:
        # the f16 forward mode has the kernels of the benchmarked dispatch: cluster forward (whhq) and fused row-wave forward (wx)
        if want("whhq"):
            if "whhq" not in out:
                out["whhq"] = torch.empty(2 * ((H + 3) // 4) * (Hp // 32) * 512, device=dev, dtype=dtype)
            call("lstm_pack_quads", whh, out["whhq"], H, Hp, F16, stream_ptr())
        if want("wihq") and _lib.load().urse_lstm_clusterx_supported(N, Np, H, Hp):
            if "wihq" not in out:
                out["wihq"] = torch.empty(2 * ((H + 3) // 4) * (Np // 32) * 512, device=dev, dtype=dtype)
            call("lstm_pack_quads_x", wih, out["wihq"], N, Np, H, F16, stream_ptr())
        if want("wx") and _lib.load().urse_lstm_rwx_supported(N, Np, H, Hp):
            if "wx" not in out:
                out["wx"] = torch.empty(2 * ((H + 15) // 16) * (Hp // 32 + Np // 32) * 4 * 512, device=dev, dtype=dtype)
            call("lstm_pack_blocks_x", wih, whh, out["wx"], N, Np, H, Hp, F16, stream_ptr())
    if dtype == torch.bfloat16 and Hp % 32 == 0:
        if want("whhq"):
            if "whhq" not in out:
                out["whhq"] = torch.empty(2 * ((H + 3) // 4) * (Hp // 32) * 512, device=dev, dtype=dtype)
            call("lstm_pack_quads", whh, out["whhq"], H, Hp, BF16, stream_ptr())
        if want("wihq") and _lib.load().urse_lstm_clusterx_supported(N, Np, H, Hp):
            if "wihq" not in out:
                out["wihq"] = torch.empty(2 * ((H + 3) // 4) * (Np // 32) * 512, device=dev, dtype=dtype)
            call("lstm_pack_quads_x", wih, out["wihq"], N, Np, H, BF16, stream_ptr())
        if want("whhb") and _lib.load().urse_lstm_wide_supported(H, Hp):
            if "whhb" not in out:
                out["whhb"] = torch.empty(2 * ((H + 15) // 16) * (Hp // 32) * 4 * 512, device=dev, dtype=dtype)
            call("lstm_pack_blocks", whh, out["whhb"], H, Hp, stream_ptr())
        if want("whhb_rw") and _lib.load().urse_lstm_rw_supported(H, Hp):
            if "whhb_rw" not in out:
                out["whhb_rw"] = torch.empty(2 * ((H + 15) // 16) * (Hp // 32) * 4 * 512, device=dev, dtype=dtype)
            call("lstm_pack_blocks_rw", whh, out["whhb_rw"], H, Hp, stream_ptr())
        if want("wx") and _lib.load().urse_lstm_rwx_supported(N, Np, H, Hp):
            if "wx" not in out:
                out["wx"] = torch.empty(2 * ((H + 15) // 16) * (Hp // 32 + Np // 32) * 4 * 512, device=dev, dtype=dtype)
            call("lstm_pack_blocks_x", wih, whh, out["wx"], N, Np, H, Hp, BF16, stream_ptr())
        if want("whhTq") and H % 8 == 0:
            C = ((H + 3) // 4 + 13) // 14
            if "whhTq" not in out:
                out["whhTq"] = torch.empty(2 * C * 4 * (H // 8) * 512, device=dev, dtype=dtype)
            call("lstm_pack_bwd_quads", whh, out["whhTq"], H, C, stream_ptr())
    return out


PACK_MULTI = os.environ.get("URSE_LSTM_PACK_MULTI", "1") != "0"     # re-pack all LSTMs of a model with one launch per layout


def lstm_pack_multi(entries, N, H, dtype, table=None):
    """re-pack LSTMs whose packed buffers exist already (entries: [(wih, whh, bih, bhh, out dict of lstm_pack)]) with ONE launch per
    layout instead of up to five per LSTM.  Returns the device pointer table (pass it back in while the buffers stay the same)."""
    first = entries[0][4]
    Np, Hp = first["Np"], first["Hp"]
    dev = entries[0][0].device
    ptr = lambda t: 0 if t is None else t.data_ptr()
    rows = [[ptr(wih), ptr(whh), ptr(bih), ptr(bhh), ptr(o["wih"]), ptr(o["wihT"]), ptr(o["bias"]), ptr(o["whh"]), ptr(o["whhT"]),
             ptr(o.get("whhq")), ptr(o.get("whhb")), ptr(o.get("wx")), ptr(o.get("wihq"))] for wih, whh, bih, bhh, o in entries]
    if table is None or table[1] != rows:
        table = (upload(torch.tensor(rows, dtype=torch.int64), dev, cached=True), rows)
    n = len(rows)
    fdt = _dt(first["wih"])
    call("lstm_pack_multi", table[0], n, N, Np, H, Hp, fdt, stream_ptr())
    if any(r[9] for r in rows):
        call("lstm_pack_quads_multi", table[0], n, H, Hp, fdt, stream_ptr())
    if any(r[10] for r in rows):
        assert fdt == BF16
        call("lstm_pack_blocks_multi", table[0], n, H, Hp, stream_ptr())
    if any(r[11] for r in rows):
        call("lstm_pack_blocks_x_multi", table[0], n, N, Np, H, Hp, fdt, stream_ptr())
    if any(r[12] for r in rows):
        call("lstm_pack_quads_x_multi", table[0], n, N, Np, H, fdt, stream_ptr())
    return table


_err_state = {}


def kernel_error_flag(device):
    """the device word the cooperative LSTM kernels (cluster / split) set when a hand-off times out."""
    st = _err_state.get(device)
    if st is None:
        st = _err_state[device] = {"flag": torch.zeros(1, device=device, dtype=torch.int32),
                                   "host": torch.zeros(1, dtype=torch.int32).pin_memory(), "event": None}
    return st["flag"]


def poll_kernel_errors(device, sync=False):
    """Fail loudly if a cooperative kernel reported a timed-out hand-off (its results are then garbage).
    sync=False (training): checks the copy requested during the PREVIOUS call and requests a new one: no host stall.
    sync=True (inference / tests): reads the flag now."""
    kernel_error_flag(device)
    st = _err_state[device]
    if sync:
        bad = int(st["flag"].item()) != 0
    else:
        bad = st["event"] is not None and st["event"].query() and int(st["host"][0]) != 0
        st["host"].copy_(st["flag"], non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record(torch.cuda.current_stream(device))
    if bad:
        raise _lib.UrseError("a cooperative LSTM kernel (cluster / split) timed out waiting for a peer workgroup: the device did "
                             "not keep all its workgroups resident; results of that step are invalid "
                             "(URSE_LSTM_CLUSTER=0 selects the streaming kernels)")


def _hout_buffer(M, ldh, H, like, dtype=None):
    """hidden-state matrix [M, ldh]: the kernels write every row's 2H columns, only the K padding needs zeros."""
    hout = torch.empty(M, ldh, device=like.device, dtype=dtype or like.dtype)
    if ldh > 2 * H:
        hout[:, 2 * H:].zero_()
    return hout


_cluster_ws = {}
USE_CLUSTER_LSTM = os.environ.get("URSE_LSTM_CLUSTER", "1") != "0"
# the cluster BPTT kernel is correct (tests/test_lstm_gpu.py) but, at 16 us + 200 KB of tile traffic per step and with
# register spills at 14 waves, still slower than the streaming BPTT (95 vs 76 ms/step at C2): opt-in until it wins
USE_CLUSTER_LSTM_BWD = os.environ.get("URSE_LSTM_CLUSTER_BWD", "0") == "1"


# CUs promised to work that is resident beside a cooperative recurrence kernel (bsrnn sets it to the workgroup target of the
# deferred weight-gradient launches while they are in flight on the second stream): the cooperative kernels' plans leave
# that many CUs alone and REFUSE a grid that would not be co-resident on the rest (-> streaming kernel) instead of spinning
CO_RESIDENT_WGS = 0
# ... and CUs left to the dynamic-mixing prefetcher's simulator kernels, which run on a low-priority stream beside the WHOLE step,
# forward included (train_se.DevicePrefetcher sets it to 8 while it stages simulated batches): at C2 the time path's cluster forward
# then forms 17 clusters of 64 sequences instead of 18 of 61 - same row tiles, same speed - and 14 CUs stay free, so that its
# workgroups never wait for a simulator workgroup to leave a CU (a 60 ms single-workgroup FFT launch once cost 15 ms per step that way)
PREFETCH_RESERVED_CUS = 0


# ... and CUs left to RCCL's all-reduce kernels while a gradient bucket is in flight (ddp.GradBucketReducer sets it at its first bucket of
# a backward pass and clears it in finish()): an all-reduce occupies up to NCCL_MAX_NCHANNELS workgroups (one per channel), which train_se /
# bench.py cap at RCCL_MAX_CHANNELS before the communicator exists.  A cooperative recurrence launched between the first bucket and finish()
# (the flow model's split BPTT; any cluster kernel) is then planned on the CUs that are left - or REFUSED and replaced by its streaming twin -
# instead of spinning on a workgroup that RCCL keeps off the chip.
COMM_RESERVED_CUS = 0
# ranks that share THIS device (bench.py / train_se set it when a non-RCCL backend puts several processes on one GPU, the gloo test path): two
# processes with a cooperative grid each - clusters that spin until the whole cluster has arrived, ~250 workgroups of 160 KB LDS - cannot be
# co-resident and would keep each other's partners off the chip until the spin bound trips.  No cooperative plan is made then (round 5
# guarded the N-split only); the streaming kernels run.
SHARED_GPU_RANKS = 1
RCCL_MAX_CHANNELS = int(os.environ.get("URSE_RCCL_MAX_CHANNELS", "32"))
COOP_REFUSALS = 0                # cooperative plans refused because of reserved CUs (diagnostic: counted, never silent)


def cap_rccl_channels():
    """call BEFORE the first collective creates the RCCL communicator: bounds the workgroups an all-reduce can occupy (a value the
    user has set stays - and is what rccl_reserved_cus() then sets aside).  Returns the effective cap."""
    os.environ.setdefault("NCCL_MAX_NCHANNELS", str(RCCL_MAX_CHANNELS))
    return rccl_reserved_cus()


def rccl_reserved_cus():
    """CUs to leave to RCCL while a bucket is in flight = the EFFECTIVE channel cap: NCCL_MAX_NCHANNELS as the communicator saw it (ADVICE r4:
    a user-set value above our default stayed in force while the reservation assumed 32 - the cooperative recurrences were then planned on
    CUs RCCL held).  Unset (cap_rccl_channels never ran) or unparsable: RCCL's own maximum of 64 channels, the safe side."""
    v = os.environ.get("NCCL_MAX_NCHANNELS")
    try:
        return max(1, int(v)) if v is not None else 64
    except ValueError:
        return 64


def reserved_cus():
    return int(CO_RESIDENT_WGS) + int(PREFETCH_RESERVED_CUS) + int(COMM_RESERVED_CUS)


class reserve_cus:
    """scoped reservation (the module globals used to stay set when an exception left the region that set them):
    with ops.reserve_cus(prefetch=8): ...   /   with ops.reserve_cus(co_resident=98): ..."""

    def __init__(self, co_resident=None, prefetch=None, comm=None):
        self.new = (co_resident, prefetch, comm)

    def __enter__(self):
        global CO_RESIDENT_WGS, PREFETCH_RESERVED_CUS, COMM_RESERVED_CUS
        self.old = (CO_RESIDENT_WGS, PREFETCH_RESERVED_CUS, COMM_RESERVED_CUS)
        co, pf, cm = self.new
        if co is not None:
            CO_RESIDENT_WGS = co
        if pf is not None:
            PREFETCH_RESERVED_CUS = pf
        if cm is not None:
            COMM_RESERVED_CUS = cm
        return self

    def __exit__(self, *exc):
        global CO_RESIDENT_WGS, PREFETCH_RESERVED_CUS, COMM_RESERVED_CUS
        CO_RESIDENT_WGS, PREFETCH_RESERVED_CUS, COMM_RESERVED_CUS = self.old
        return False


def _note_refusal(fits_without_reservation):
    """a cooperative plan failed: count it as a refusal if the reservation (not the shape) is why."""
    global COOP_REFUSALS
    if reserved_cus() > 0 and fits_without_reservation():
        COOP_REFUSALS += 1


def lstm_cluster_plan(H, Hp, n_seq):
    """None if the persistent cluster kernel does not support this shape."""
    import ctypes
    if SHARED_GPU_RANKS > 1:
        return None
    plan = (ctypes.c_int64 * 6)()
    lib = _lib.load()
    if lib.urse_lstm_cluster_plan(H, Hp, n_seq, reserved_cus(), plan) != 0:
        _note_refusal(lambda: lib.urse_lstm_cluster_plan(H, Hp, n_seq, 0, plan) == 0)
        return None
    return list(plan)


# clusters of the time path's forward formed from workgroups on ONE XCD (the h all-gather then stays in that XCD's L2)
CLUSTER_XCD_AWARE = os.environ.get("URSE_LSTM_CLUSTER_XCD", "1") != "0"


def lstm_fwd_cluster(gx, whhq, H, Hp, n_seq, seq_len, inner, outer, stride, save=True, xcd_aware=None, bf16_copy=False):
    """persistent cluster LSTM forward (bf16 | f16 operands): see csrc/lstm_cluster.hip.  With save the gx buffer comes back holding the gate
    ACTIVATIONS in bf16 (also under f16 operands: read it as gx.view(torch.bfloat16)).  bf16_copy (f16): -> (hout, c, err, hout_bf16)."""
    xcd_aware = CLUSTER_XCD_AWARE if xcd_aware is None else xcd_aware
    plan = lstm_cluster_plan(H, Hp, n_seq)
    M, dev = gx.shape[0], gx.device
    key = (dev, H, Hp, n_seq, plan[4], plan[5])
    if key not in _cluster_ws:
        _cluster_ws[key] = (torch.zeros(plan[4], device=dev, dtype=torch.bfloat16),
                            torch.zeros(plan[5], device=dev, dtype=torch.int32),
                            kernel_error_flag(dev))
    hx, cnt, err = _cluster_ws[key]
    ldh = kpad(2 * H, gx.dtype)
    hout = _hout_buffer(M, ldh, H, gx)
    hout2 = _hout_buffer(M, ldh, H, gx, torch.bfloat16) if (bf16_copy and gx.dtype == torch.float16) else None
    c = torch.empty(M, 2 * H, device=dev, dtype=torch.float32) if save else None
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_cluster_fwd", gx, gx.stride(0), whhq, hout, ldh,
               c, hx, cnt, err, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), reserved_cus(), int(bool(xcd_aware)), _dt(gx), hout2,
               stream_ptr())
    return (hout, c, err, hout2 if hout2 is not None else hout) if bf16_copy else (hout, c, err)


# the cluster forward with the input projection fused (csrc/lstm_clusterx.hip, round 5): no gate-projection GEMM on the time path, no gx matrix
USE_CLUSTERX_LSTM = os.environ.get("URSE_LSTM_CLUSTERX", "1") != "0"


def lstm_clusterx_supported(N, Np, H, Hp):
    return bool(_lib.load().urse_lstm_clusterx_supported(N, Np, H, Hp))


# the band path through the same kernel in ROUNDS (round 6): every co-resident cluster keeps its weights and takes 64 sequences per round
BAND_CLUSTERX = os.environ.get("URSE_LSTM_BAND_CLUSTERX", "1") != "0"
# ... and the time path above 1,152 sequences per direction (batches of more than 33 utterances at 48 kHz)
TIME_CLUSTERX_ROUNDS = os.environ.get("URSE_LSTM_TIME_CLUSTERX_ROUNDS", "1") != "0"


def lstm_clusterx_plan(H, Hp, n_seq):
    """[C, clusters per direction, sequences per cluster, rows_pad, hx elements, counters, rounds] of the fused cluster forward, or None (unsupported
    shape, ranks sharing the GPU, or the reservation leaves no room for a cluster).  rounds > 1: more sequences than the clusters hold at once."""
    import ctypes
    if SHARED_GPU_RANKS > 1:
        return None
    plan = (ctypes.c_int64 * 7)()
    lib = _lib.load()
    if lib.urse_lstm_clusterx_plan(H, Hp, n_seq, reserved_cus(), plan) != 0:
        _note_refusal(lambda: lib.urse_lstm_clusterx_plan(H, Hp, n_seq, 0, plan) == 0)
        return None
    return list(plan)


# what the two band-path forwards cost per launch, each alone at T = 401, K = 34 (profiles/r06_exp_band_clusterx_v2.log): the cluster kernel 6.3 us per step whatever
# a round carries (B 32: 12 rounds 2.62 ms, B 8: 3 rounds 0.70 ms, B 4: 2 rounds 0.46 ms) + ~50 us of launch prologue; the row-wave kernel 3.5 ns per
# sequence and step (B 32: 3.07 ms) with a floor of 36 us per step (B 8: 1.38 ms, B 4: 1.23 ms - a workgroup's pass over the weights per step)
CLUSTERX_US_PER_STEP, CLUSTERX_US_PROLOGUE, RWX_NS_PER_SEQ_STEP, RWX_US_PER_STEP_FLOOR = 6.3, 50.0, 3.5, 36.0


def band_clusterx_pays(H, Hp, n_seq, seq_len=34):
    """True where the plan exists, needs more than one round (one round is lstm_cluster_plan's business), and prices at least 5 % below the row-wave
    kernel (C2 without a reservation: 12 rounds, 2.6 against 3.1 ms; beside 32 reserved CUs: 15 clusters, 14 rounds - the row-wave kernel keeps the launch)."""
    plan = lstm_clusterx_plan(H, Hp, n_seq)
    if plan is None or plan[6] <= 1:
        return False
    cx = plan[6] * seq_len * CLUSTERX_US_PER_STEP + CLUSTERX_US_PROLOGUE
    rwx = seq_len * max(RWX_US_PER_STEP_FLOOR, 2 * n_seq * RWX_NS_PER_SEQ_STEP * 1e-3)
    return cx < 0.95 * rwx


def lstm_fwd_clusterx(xn, wihq, whhq, bias, N, H, Hp, n_seq, seq_len, inner, outer, stride, save=True, xcd_aware=None, bf16_copy=False):
    """cluster LSTM forward with the input projection fused (bf16 | f16 operands): xn [M, Np] -> (gates [M, 8H] bf16 activations or None, hout, c, err);
    bf16_copy (f16): one more element, hout_bf16.  Any n_seq lstm_clusterx_plan accepts (in rounds above clusters * 64 sequences per direction)."""
    xcd_aware = CLUSTER_XCD_AWARE if xcd_aware is None else xcd_aware
    plan = lstm_clusterx_plan(H, Hp, n_seq)
    if plan is None:
        raise RuntimeError("lstm_fwd_clusterx: no cluster plan for H=%d n_seq=%d" % (H, n_seq))
    M, Np, dev = xn.shape[0], xn.shape[1], xn.device
    n_hx = int(plan[4])
    key = ("x", dev, H, Hp, n_seq, n_hx, plan[5])
    if key not in _cluster_ws:
        _cluster_ws[key] = (torch.zeros(n_hx, device=dev, dtype=torch.bfloat16),
                            torch.zeros(plan[5], device=dev, dtype=torch.int32),
                            kernel_error_flag(dev))
    hx, cnt, err = _cluster_ws[key]
    ldh = kpad(2 * H, xn.dtype)
    hout = _hout_buffer(M, ldh, H, xn)
    hout2 = _hout_buffer(M, ldh, H, xn, torch.bfloat16) if (bf16_copy and xn.dtype == torch.float16 and save) else None
    c = torch.empty(M, 2 * H, device=dev, dtype=torch.float32) if save else None
    gates = torch.empty(M, 8 * H, device=dev, dtype=torch.bfloat16) if save else None
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_clusterx_fwd", xn, xn.stride(0), wihq, bias, whhq, gates, 8 * H, hout, ldh, c,
               hx, cnt, err, N, Np, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), reserved_cus(), int(bool(xcd_aware)), _dt(xn), hout2,
               stream_ptr())
    if bf16_copy:
        return gates, hout, c, err, (hout2 if hout2 is not None else hout)
    return gates, hout, c, err


# which hidden sizes take the generalised cluster kernel: "768" by default (H = 392 keeps lstm_cluster.hip unless asked)
CLUSTER2_H = tuple(int(v) for v in os.environ.get("URSE_LSTM_CLUSTER2_H", "768").split(",") if v)


def lstm_cluster2_plan(H, Hp, n_seq):
    import ctypes
    if SHARED_GPU_RANKS > 1:
        return None
    plan = (ctypes.c_int64 * 4)()
    if _lib.load().urse_lstm_cluster2_plan(H, Hp, n_seq, reserved_cus(), plan) != 0:
        _note_refusal(lambda: _lib.load().urse_lstm_cluster2_plan(H, Hp, n_seq, 0, plan) == 0)
        return None
    return list(plan)


def lstm_cluster2_chunks(H, Hp, n_seq, seq_len, inner, outer, stride):
    """[(first sequence, count)] launches of the generalised cluster kernel that cover n_seq sequences, or None.  One launch takes
    what the clusters resident on the chip hold (H = 768: 5 clusters x 64 sequences per direction); more sequences run as
    several launches when every sequence is a contiguous block of rows (the band path: inner 1, stride 1) - at H = 768 a launch
    is 48 steps x 7 us against 43 us per step of the streaming kernel, which re-reads 4.7 MB of weights per step and CU."""
    if lstm_cluster2_plan(H, Hp, n_seq) is not None:
        return [(0, n_seq)]
    if not (inner == 1 and stride == 1 and outer == seq_len):
        return None
    probe = lstm_cluster2_plan(H, Hp, 1)
    if probe is None:
        return None
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    cap = max(1, (cus - reserved_cus() - 4) // 2 // probe[0]) * 64
    n = (n_seq + cap - 1) // cap
    if n > CLUSTER2_MAX_CHUNKS:
        return None
    per = (n_seq + n - 1) // n
    return [(s0, min(per, n_seq - s0)) for s0 in range(0, n_seq, per)]


CLUSTER2_MAX_CHUNKS = int(os.environ.get("URSE_LSTM_CLUSTER2_MAX_CHUNKS", "6"))


def lstm_fwd_cluster2(gx, whhq, H, Hp, n_seq, seq_len, inner, outer, stride, save=True):
    """generalised persistent cluster LSTM forward (bf16 | f16 operands, by gx.dtype): see csrc/lstm_cluster2.hip."""
    chunks = lstm_cluster2_chunks(H, Hp, n_seq, seq_len, inner, outer, stride)
    M, dev = gx.shape[0], gx.device
    ldh = kpad(2 * H, gx.dtype)
    hout = _hout_buffer(M, ldh, H, gx)
    c = torch.empty(M, 2 * H, device=dev, dtype=torch.float32) if save else None
    err = None
    for s0, n in chunks:
        plan = lstm_cluster2_plan(H, Hp, n)
        key = ("c2", dev, H, Hp, n, plan[3])
        if key not in _cluster_ws:
            _cluster_ws[key] = (torch.empty(plan[3], device=dev, dtype=torch.bfloat16), kernel_error_flag(dev))
        hx, err = _cluster_ws[key]
        if len(chunks) == 1:
            g_, h_, c_ = gx, hout, c
        else:                                   # sequence s is rows [s * seq_len, (s + 1) * seq_len)
            r0, r1 = s0 * seq_len, (s0 + n) * seq_len
            g_, h_, c_ = gx[r0:r1], hout[r0:r1], (c[r0:r1] if save else None)
        timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_cluster2_fwd", g_, gx.stride(0), whhq, h_, ldh, c_, hx,
                   err, H, Hp, n, seq_len, inner, outer, stride, int(save), reserved_cus(), _dt(gx), stream_ptr())
    return hout, c, err


def lstm_fwd(gx, whh, H, Hp, n_seq, seq_len, inner, outer, stride, save=True, rows16=0, bf16_copy=False):
    """gx [M, 8H] (overwritten by gate activations if save - in bf16 also under f16 operands) -> (hout [M, kpad(2H)], c [M, 2H] f32);
    bf16_copy (f16): -> (hout, c, hout_bf16)."""
    M = gx.shape[0]
    ldh = kpad(2 * H, gx.dtype)
    hout = _hout_buffer(M, ldh, H, gx)
    hout2 = _hout_buffer(M, ldh, H, gx, torch.bfloat16) if (bf16_copy and gx.dtype == torch.float16) else None
    c = torch.empty(M, 2 * H, device=gx.device, dtype=torch.float32) if save else None
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_bidir_fwd", gx, gx.stride(0), whh, hout, ldh,
               c, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), _dt(gx), rows16, hout2, stream_ptr())
    return (hout, c, hout2 if hout2 is not None else hout) if bf16_copy else (hout, c)


USE_WIDE_LSTM = os.environ.get("URSE_LSTM_WIDE", "1") != "0"
# GroupNorm statistics of a dual-path output from the epilogue of the fc GEMM that produces it (else a separate pass)
FUSE_GN_STATS = os.environ.get("URSE_FUSE_GN_STATS", "1") != "0"
# run the dual-path weight-gradient GEMMs on a second stream beside the time path's BPTT kernel (which fills 136 CUs)
TN_OVERLAP = os.environ.get("URSE_TN_OVERLAP", "1") != "0"
def low_priority_stream(device, force=False):
    """The stream of the deferred weight-gradient GEMMs (and, with force=True, of the dynamic-mixing prefetcher: its simulator
    workgroups must never keep a cooperative recurrence workgroup off its CU).  URSE_SIDE_STREAM_PRIORITY=low makes it a HIP stream of the LOWEST
    priority (torch only offers normal / high), so that its work only takes CUs the compute stream leaves idle; measured
    WORSE than equal priority (196 vs 189 ms/step): the band path's BPTT runs alone (30 -> 25 ms/step) but all the GEMM work
    then piles up beside the time path's BPTT (48 -> 58 ms/step).  Default: normal priority."""
    if not force and os.environ.get("URSE_SIDE_STREAM_PRIORITY", "normal") != "low":
        return torch.cuda.Stream(device=device)
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        least, greatest = ctypes.c_int(), ctypes.c_int()
        st = ctypes.c_void_p()
        with torch.cuda.device(device):
            if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
                raise RuntimeError("priority range")
            if hip.hipStreamCreateWithPriority(ctypes.byref(st), ctypes.c_uint(1), ctypes.c_int(least.value)) != 0:   # 1 = non-blocking
                raise RuntimeError("stream create")
        return torch.cuda.ExternalStream(st.value, device=device)
    except Exception:
        return torch.cuda.Stream(device=device)


# workgroups the TN launches of the second queue are sized to (bsrnn._run_deferred_wgrads): 120 = the CUs the time path's
# BPTT (136 workgroups) leaves idle; beside the band path's BPTT, which fills the chip, four R-slices (84 workgroups) are
# the best trade (same-box: 180.9 ms/step at 120, 175.3 at 84, 186 at 63; measured with the faster TN kernel)
TN_SHADOW_WGS = int(os.environ.get("URSE_TN_SHADOW_WGS", "98"))
# ... beside the N-split BPTT (136 workgroups of 16 waves): 134.3 / 134.3 / 133.3 / 137.7 ms per step at 84 / 98 / 112 / 126 (profiles/r04_ab_tn_shadow_v1.log;
# 112 against 98 alone, order reversed: inside the spread, r04_ab_order_bias_v1.log).
# Its own switch: the flow model's cooperative split BPTT plans its grid on the CUs this number leaves, and at 112 it no longer fit (its train step went
# from 87.8 to 104.3 ms when the one number served both)
TN_SHADOW_WGS_NSPLIT = int(os.environ.get("URSE_TN_SHADOW_WGS_NSPLIT", "112"))
TN_SHADOW_WGS_BAND = int(os.environ.get("URSE_TN_SHADOW_WGS_BAND", "84"))
# batches the join behind the time path's BPTT leaves running (their operands stay alive that much longer)
TN_JOIN_LAG = int(os.environ.get("URSE_TN_JOIN_LAG", "2"))   # same-box: 176.7 (0), 175.3 (1), 174.4 (2), 174.3 (4) ms/step
# how many of the deferred launches (3 per half layer: fc, dir 0, dir 1) start beside the band path's BPTT; the rest wait
# for the time path's, which leaves 120 CUs idle (unset / negative = all; 2 or 1 measured 182-185 vs 174 ms/step)
TN_BAND_PARTS = int(os.environ.get("URSE_TN_BAND_PARTS", "-1"))
TN_BAND_PARTS = None if TN_BAND_PARTS < 0 else TN_BAND_PARTS
TN_OVERLAP_TAIL = os.environ.get("URSE_TN_OVERLAP_TAIL", "0") != "0"     # the last half layer's weight gradients beside the band split's backward (measured: 132.95 / 134.61 ms per step off, 135.01 / 134.19 on - no gain, off)
# the mask decoder's grouped weight gradients on the second queue: measured neutral once the A/B's order bias was removed (profiles/r04_ab_order_bias_v1.log),
# it delays the 'md' bucket of the DDP reducer by TN_JOIN_LAG joins and its launch is not sized to the CUs the second queue declares: off
DEFER_MASKDEC_WGRADS = os.environ.get("URSE_DEFER_MASKDEC_WGRADS", "0") != "0"
TN_OVERLAP_BAND = os.environ.get("URSE_TN_OVERLAP_BAND", "1") != "0"   # also start deferred wgrads beside the band path's BPTT
# the wide kernel wins once there are enough 64-sequence workgroups to fill the chip in both directions
WIDE_MIN_SEQ = int(os.environ.get("URSE_LSTM_WIDE_MIN_SEQ", str(64 * 128)))
# the cluster forward takes any path whose sequences fit its 18 x 64 capacity; at C2 that is the time path only (the band path
# has 12,832).  Tests of the C2 kernel set on small batches set this so that the band path skips it as it does at C2.
BAND_PATH_NO_CLUSTER = os.environ.get("URSE_LSTM_BAND_NO_CLUSTER", "0") == "1"


def lstm_fwd_wide(gx, whhb, H, Hp, n_seq, seq_len, inner, outer, stride, save=True):
    """wide streaming LSTM forward (bf16, 64 sequences per workgroup): see csrc/lstm_wide.hip."""
    M = gx.shape[0]
    ldh = kpad(2 * H, gx.dtype)
    hout = _hout_buffer(M, ldh, H, gx)
    c = torch.empty(M, 2 * H, device=gx.device, dtype=torch.float32)
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_wide_fwd", gx, gx.stride(0), whhb, hout, ldh,
               c, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), stream_ptr())
    return hout, (c if save else None)


# the row-wave kernel (csrc/lstm_rw.hip) takes the band path once there are enough 16-sequence tiles to give every CU a
# workgroup of 6-7 of them (C2: 802 tiles per direction); below that the wide / streaming kernels keep it
USE_RW_LSTM = os.environ.get("URSE_LSTM_RW", "1") != "0"
RW_MIN_SEQ = int(os.environ.get("URSE_LSTM_RW_MIN_SEQ", str(16 * 6 * 64)))


def lstm_rw_supported(H, Hp):
    return bool(_lib.load().urse_lstm_rw_supported(H, Hp))


# the kernel form with two adjacent units per lane and block pair (wider, contiguous accesses); its weights come from lstm_pack_blocks_rw
RW_PAIRED = os.environ.get("URSE_LSTM_RW_PAIRED", "0") == "1"      # (round 6: the paired form exists in variant builds only, -DURSE_EXPERIMENTS)


def lstm_fwd_rw(gx, whhb, H, Hp, n_seq, seq_len, inner, outer, stride, save=True, target_wgs=0, paired=False):
    """row-wave LSTM forward (bf16, 16 sequences per wave, shared LDS weight ring): see csrc/lstm_rw.hip.
    paired: whhb is the column-permuted packing of lstm_pack_blocks_rw."""
    M = gx.shape[0]
    ldh = kpad(2 * H, gx.dtype)
    hout = _hout_buffer(M, ldh, H, gx)
    c = torch.empty(M, 2 * H, device=gx.device, dtype=torch.float32)
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_rw_fwd", gx, gx.stride(0), whhb, hout, ldh,
               c, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), int(target_wgs), int(paired), stream_ptr())
    return hout, (c if save else None)


# the fused form (input projection inside the recurrence, csrc/lstm_rwx.hip): no gate-projection GEMM, no gx matrix
USE_RWX_LSTM = os.environ.get("URSE_LSTM_RWX", "1") != "0"


def lstm_rwx_supported(N, Np, H, Hp):
    return bool(_lib.load().urse_lstm_rwx_supported(N, Np, H, Hp))


def lstm_fwd_rwx(xn, wx, bias, N, H, Hp, n_seq, seq_len, inner, outer, stride, save=True, target_wgs=0, bf16_copy=False):
    """row-wave LSTM forward with the input projection fused: xn [M, Np] bf16 | f16 -> (gates [M, 8H] bf16 activations or None, hout, c);
    bf16_copy (f16): -> (gates, hout, c, hout_bf16)."""
    M, Np = xn.shape[0], xn.shape[1]
    ldh = kpad(2 * H, xn.dtype)
    hout = _hout_buffer(M, ldh, H, xn)
    hout2 = _hout_buffer(M, ldh, H, xn, torch.bfloat16) if (bf16_copy and xn.dtype == torch.float16 and save) else None
    c = torch.empty(M, 2 * H, device=xn.device, dtype=torch.float32)
    gates = torch.empty(M, 8 * H, device=xn.device, dtype=torch.bfloat16) if save else None
    timed_call("lstm_fwd_time" if stride > 1 else "lstm_fwd_band", "lstm_rwx_fwd", xn, xn.stride(0), wx, bias, gates, 8 * H, hout, ldh, c,
               N, Np, H, Hp, n_seq, seq_len, inner, outer, stride, int(save), int(target_wgs), _dt(xn), hout2, stream_ptr())
    if bf16_copy:
        return gates, hout, (c if save else None), (hout2 if hout2 is not None else hout)
    return gates, hout, (c if save else None)


def lstm_bwd_cluster(dh, gates, c, whhTq, H, Hp, n_seq, seq_len, inner, outer, stride):
    """persistent cluster BPTT (bf16): gates (saved activations) is overwritten with d(pre-activations)."""
    plan = lstm_cluster_plan(H, Hp, n_seq)
    dev = gates.device
    key = ("bwd", dev, H, Hp, n_seq, plan[1], plan[5])
    if key not in _cluster_ws:
        _cluster_ws[key] = (torch.zeros(2 * 2 * plan[1] * 64 * 4 * H, device=dev, dtype=torch.bfloat16),
                            torch.zeros(plan[5], device=dev, dtype=torch.int32),
                            kernel_error_flag(dev))
    dgx, cnt, err = _cluster_ws[key]
    timed_call("lstm_bwd_time" if stride > 1 else "lstm_bwd_band", "lstm_cluster_bwd", dh, dh.stride(0), gates,
               gates.stride(0), c, whhTq, dgx, cnt, err, H, Hp, n_seq, seq_len, inner, outer, stride, reserved_cus(), stream_ptr())
    return gates, err


# split BPTT (csrc/lstm_split.hip): correct (tests/test_lstm_gpu.py) but not faster than the streaming kernel yet (7.0-8.9 vs
# 7.2 ms per time-path launch): the serial phases of a step (hand-off wait, dgates, barrier chain, publish) cost 8 us with
# the weight stream switched off (scripts/abl_lstm.py).  Opt-in until it wins.
USE_SPLIT_LSTM_BWD = os.environ.get("URSE_LSTM_SPLIT_BWD", "0") == "1"
# ... except where few sequences meet a big hidden size (flow model, H = 768, 48-96 time-path sequences: the streaming kernel
# keeps 3-6 workgroups busy at 67 us per step): there the 16-row, up-to-6-way split is the default
SPLIT_BWD_MIN_H = int(os.environ.get("URSE_LSTM_SPLIT_BWD_MIN_H", "512"))


def lstm_split_plan(H, n_seq):
    """None if the split BPTT kernel does not support this shape."""
    import ctypes
    if SHARED_GPU_RANKS > 1:
        return None
    plan = (ctypes.c_int64 * 4)()
    if _lib.load().urse_lstm_split_plan(H, n_seq, reserved_cus(), plan) != 0:
        _note_refusal(lambda: _lib.load().urse_lstm_split_plan(H, n_seq, 0, plan) == 0)
        return None
    return list(plan)


# row-block launches of the split BPTT for the flow model's band path: measured neutral against the streaming kernel (train step
# 114.5 ms either way, profiles/r02_ab_flow_split_chunks_v1.log), so one launch or none by default
SPLIT_BWD_MAX_CHUNKS = int(os.environ.get("URSE_LSTM_SPLIT_BWD_MAX_CHUNKS", "1"))


def lstm_split_chunks(H, n_seq, seq_len, inner, outer, stride):
    """[(first sequence, count)] launches of the split BPTT that cover n_seq sequences, or None: one launch when all its
    workgroups fit the chip, else (band path only: every sequence a contiguous block of rows) the fewest equal row blocks that do."""
    if lstm_split_plan(H, n_seq) is not None:
        return [(0, n_seq)]
    if not (inner == 1 and stride == 1 and outer == seq_len):
        return None
    for n in range(2, SPLIT_BWD_MAX_CHUNKS + 1):
        per = (n_seq + n - 1) // n
        if lstm_split_plan(H, per) is not None:
            return [(s0, min(per, n_seq - s0)) for s0 in range(0, n_seq, per)]
    return None


def lstm_bwd_split(dh, gates, c, whhT, H, n_seq, seq_len, inner, outer, stride):
    """split BPTT (bf16): gates (saved activations) is overwritten with d(pre-activations)."""
    chunks = lstm_split_chunks(H, n_seq, seq_len, inner, outer, stride)
    dev = gates.device
    err = None
    for s0, n in chunks:
        plan = lstm_split_plan(H, n)
        key = ("split", dev, H, n, plan[2])
        if key not in _cluster_ws:
            _cluster_ws[key] = (torch.empty(plan[2], device=dev, dtype=torch.float32), kernel_error_flag(dev))
        xbuf, err = _cluster_ws[key]
        if len(chunks) == 1:
            d_, g_, c_ = dh, gates, c
        else:                                   # sequence s is rows [s * seq_len, (s + 1) * seq_len)
            r0, r1 = s0 * seq_len, (s0 + n) * seq_len
            d_, g_, c_ = dh[r0:r1], gates[r0:r1], c[r0:r1]
        timed_call("lstm_bwd_time" if stride > 1 else "lstm_bwd_band", "lstm_split_bwd", d_, dh.stride(0), g_, gates.stride(0), c_,
                   whhT, xbuf, err, H, n, seq_len, inner, outer, stride, reserved_cus(), stream_ptr())
    return gates, err


# N-split BPTT (csrc/lstm_nsplit.hip): pairs of workgroups split the output columns of the recurrent product - half the weight stream per
# step and CU; the time path at C2 (1,088 sequences -> 136 workgroups, as many as the streaming kernel uses)
USE_NSPLIT_LSTM_BWD = os.environ.get("URSE_LSTM_NSPLIT_BWD", "1") != "0"
NSPLIT_MAX_SEQ = int(os.environ.get("URSE_LSTM_NSPLIT_MAX_SEQ", "2304"))       # beyond that the 32-sequence streaming geometry has the rows it needs


# (SHARED_GPU_RANKS: defined beside the CU reservations above; the N-split is not planned on a shared GPU either)


def _nsplit_reserved():
    """CUs the N-split plan must leave alone.  Its workgroups wait for ONE partner each, not for the whole grid: a member whose partner has
    not started yet spins until any other workgroup on the chip retires.  The second queue's weight-gradient GEMMs are finite work SIZED beside
    this launch (TN_SHADOW_WGS_NSPLIT), so they only delay a pair and are not reserved for; RCCL's all-reduce kernels and the prefetcher's
    simulator kernels are declared by others and honoured here: the grid must fit beside them."""
    return int(COMM_RESERVED_CUS) + int(PREFETCH_RESERVED_CUS)


def lstm_nsplit_plan(H, n_seq):
    import ctypes
    if SHARED_GPU_RANKS > 1:
        return None
    plan = (ctypes.c_int64 * 3)()
    if _lib.load().urse_lstm_nsplit_plan(H, n_seq, _nsplit_reserved(), plan) != 0:
        return None
    return list(plan)


def use_nsplit_bwd(H, Hp, dt, path, sm, has_whhTq=False):
    """does this half layer's BPTT run on the N-split kernel (bsrnn.dualpath_bwd's dispatch: after the opt-in cluster BPTT and the split BPTT of
    big hidden sizes, before the streaming kernel)?  Only where the library has an N-split kernel for H (392)."""
    return (not (USE_CLUSTER_LSTM_BWD and has_whhTq and lstm_cluster_plan(H, Hp, sm["n_seq"]) is not None) and
            not ((USE_SPLIT_LSTM_BWD or H >= SPLIT_BWD_MIN_H) and dt == torch.bfloat16 and lstm_split_chunks(H, **sm) is not None) and
            USE_NSPLIT_LSTM_BWD and dt == torch.bfloat16 and path not in BWD_ROWS16 and sm["n_seq"] <= NSPLIT_MAX_SEQ and
            sm["n_seq"] * sm["seq_len"] >= 4096 and lstm_nsplit_plan(H, sm["n_seq"]) is not None)


def wgrad_shadow_wgs(path, nsplit):
    """workgroups the second queue's weight-gradient launches are sized to beside this half layer's BPTT.  TN_SHADOW_WGS_NSPLIT applies ONLY
    beside the N-split kernel: every other time-path BPTT - the flow model's cooperative split BPTT among them, whose plan is made on the CUs
    this number leaves - keeps TN_SHADOW_WGS (one number for both cost the flow train step 87.8 -> 104.3 ms in round 4; tests/test_host_cpu.py)."""
    if path != "t":
        return TN_SHADOW_WGS_BAND
    return TN_SHADOW_WGS_NSPLIT if nsplit else TN_SHADOW_WGS


def lstm_bwd_nsplit(dh, gates, c, whhT, H, n_seq, seq_len, inner, outer, stride):
    """N-split BPTT (bf16): gates (saved activations) is overwritten with d(pre-activations)."""
    plan = lstm_nsplit_plan(H, n_seq)
    dev = gates.device
    key = ("nsplit", dev, plan[2])
    if key not in _cluster_ws:
        _cluster_ws[key] = (torch.zeros(plan[2], device=dev, dtype=torch.int32), kernel_error_flag(dev))
    flags, err = _cluster_ws[key]
    timed_call("lstm_bwd_time" if stride > 1 else "lstm_bwd_band", "lstm_nsplit_bwd", dh, dh.stride(0), gates, gates.stride(0), c, whhT,
               flags, err, H, n_seq, seq_len, inner, outer, stride, _nsplit_reserved(), stream_ptr())
    return gates, err


# geometry override of the streaming BPTT per path ("t" / "f"): 0 = the library decides from the number of sequences
# (32-sequence workgroups of 8 waves once there are >= 4096 of them); 2 | 16 forces that geometry - the parity tests of
# the benchmarked kernel set use it to run the C2 kernels on batches a CPU oracle can follow
BWD_ROWS16 = {k: int(os.environ[e]) for k, e in (("t", "URSE_BWD_ROWS_T"), ("f", "URSE_BWD_ROWS_F")) if e in os.environ}


# Band path (many short sequences, 32 per workgroup): 2 * ceil(n_seq / 32) workgroups rarely fill whole rounds of the 256 CUs (C2: 802 =
# 3.13 rounds, the last one 13 % full).  With this switch the sequences of the incomplete round run as 16-sequence workgroups in a
# second launch on another stream AT THE SAME TIME (their rows are independent; sequence s of a band-path layout is rows [s * K, (s + 1) * K),
# so a sub-range is a pointer offset): alone 3.95 -> 3.67 ms per launch, bit-identical (scripts/exp_band_tail.py).  In the train step it
# gains nothing (same-box 160.5 / 161.9 without, 161.4 - 163.6 with, profiles/r03_ab_band_tail_v1.log): the weight-gradient GEMMs of
# the second queue already run in that last round.  Off.
BAND_TAIL_SPLIT = os.environ.get("URSE_BAND_TAIL_SPLIT", "0") == "1"
BAND_TAIL_EXTRA = int(os.environ.get("URSE_BAND_TAIL_EXTRA", "16"))      # 32-sequence workgroups per direction moved to the tail launch besides
_tail_streams = {}


def lstm_bwd(dh, gates, c, whhT, H, n_seq, seq_len, inner, outer, stride, rows16=0):
    """gates (saved activations) is overwritten with d(pre-activations)."""
    tname = "lstm_bwd_time" if stride > 1 else "lstm_bwd_band"
    if BAND_TAIL_SPLIT and gates.is_cuda and rows16 == 0 and inner == 1 and stride == 1 and outer == seq_len and n_seq >= 32 * 128 \
            and gates.dtype == torch.bfloat16:
        per_dir = -(-n_seq // 32)
        full = (2 * per_dir // 256) * 128 - BAND_TAIL_EXTRA                  # workgroups per direction in whole rounds
        tail = per_dir - full
        if full > 0 and 0 < tail * 2 <= 192:
            dev = gates.device
            side = _tail_streams.get(dev)
            if side is None:
                side = _tail_streams[dev] = torch.cuda.Stream(device=dev)
            n_main = full * 32
            r0 = n_main * seq_len
            main = torch.cuda.current_stream(dev)
            fork = torch.cuda.Event()
            fork.record(main)
            side.wait_event(fork)
            timed_call(tname, "lstm_bidir_bwd", dh, dh.stride(0), gates, gates.stride(0), c, whhT, H, n_main, seq_len, inner, outer, stride,
                       _dt(gates), 0, stream_ptr())
            with torch.cuda.stream(side):
                call("lstm_bidir_bwd", dh[r0:], dh.stride(0), gates[r0:], gates.stride(0), c[r0:], whhT, H, n_seq - n_main, seq_len, inner,
                     outer, stride, _dt(gates), 1, stream_ptr())
                join = torch.cuda.Event()
                join.record(side)
            main.wait_event(join)
            return gates
    timed_call(tname, "lstm_bidir_bwd", dh, dh.stride(0), gates,
               gates.stride(0), c, whhT, H, n_seq, seq_len, inner, outer, stride, _dt(gates), rows16, stream_ptr())
    return gates


# ---------------------------------------------------------------------------------------------
# losses / optimizer
# ---------------------------------------------------------------------------------------------
import ctypes as _ct


class _MRL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, estimate, target, windows, eps, td_weight, nan_guard=False):
        B, L = estimate.shape
        dev = estimate.device
        loss = torch.empty(B, device=dev, dtype=torch.float32)
        sums = torch.empty(B * 5, device=dev, dtype=torch.float64)
        acc = torch.empty(B * 2, device=dev, dtype=torch.float64)
        need = estimate.requires_grad
        G = torch.empty(B, L, device=dev, dtype=torch.float32) if need else None
        warr = (_ct.c_int32 * len(windows))(*windows)
        call("mrl1_loss_fwd", target, estimate, loss, G, sums, acc, B, L, warr, len(windows), float(eps),
             float(td_weight), stream_ptr())
        ctx.saved = (estimate, target, G, sums, eps, loss if nan_guard else None)
        return loss

    @staticmethod
    def backward(ctx, gl):
        estimate, target, G, sums, eps, loss = ctx.saved
        B, L = estimate.shape
        de = torch.empty_like(estimate)
        c1 = torch.empty(B, device=estimate.device, dtype=torch.float64)
        if loss is None:
            call("mrl1_loss_bwd", target, estimate, G, sums, _f32c(gl), de, c1, B, L, float(eps), stream_ptr())
        else:
            call("mrl1_loss_bwd_guarded", target, estimate, G, sums, _f32c(gl), loss, de, c1, B, L, float(eps), stream_ptr())
        ctx.saved = None
        return de, None, None, None, None, None


def mr_l1_loss(target, estimate, window_sz=(256, 512, 768, 1024), eps=1e-6, time_domain_weight=0.5, nan_guard=False):
    """espnet2 MultiResL1SpecLoss(normalize_variance=True, reduction='sum').forward(target, estimate) -> [B].
    nan_guard: the reference's NaN-loss skip (d_model.py:75-77) without a host sync: if any loss[b] is NaN the gradient
    of the whole batch is zero (the step then runs on zero gradients, as `se_speech.mean() * 0` makes it there)."""
    require_cuda(target, estimate)
    return _MRL1.apply(_f32c(estimate), _f32c(target), tuple(int(w) for w in window_sz), eps, time_domain_weight,
                       bool(nan_guard))


def si_snr_loss(ref, inf):
    """espnet2 SISNRLoss()(ref, inf) -> [B] (no gradient: the reference only logs it, d_model.py:79-80)."""
    require_cuda(ref, inf)
    ref, inf = _f32c(ref.detach()), _f32c(inf.detach())
    B, L = ref.shape
    loss = torch.empty(B, device=ref.device, dtype=torch.float32)
    sums = torch.empty(B * 5, device=ref.device, dtype=torch.float64)
    call("sisnr_fwd", ref, inf, loss, sums, B, L, stream_ptr())
    return loss


class FusedClipAdamW:
    """clip_grad_norm_(max_norm) + torch.optim.AdamW semantics on flat f32 buffers, one kernel.

    With ``core`` (a BSRNNCore): parameters that got no gradient this step (the bands above fs/2; on several ranks: the
    bands NO rank used) are left alone exactly as torch.optim.AdamW leaves parameters whose ``.grad`` is None - no
    weight decay, no moment decay, no step count (each band keeps its own) - and an update whose gradients came out of a
    cooperative LSTM kernel that reported a timed-out hand-off is skipped like one with a non-finite norm."""

    def __init__(self, flat_params, flat_grads, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6,
                 max_norm=0.5, core=None):
        require_cuda(flat_params, flat_grads)
        n = flat_params.numel()
        self.p, self.g = flat_params, flat_grads[:n]
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.step_count = 0
        self.normsq = torch.zeros(1, device=flat_params.device, dtype=torch.float64)
        self.core = core
        if core is not None:
            dev = flat_params.device
            self.slot, self.used, self.n_slot = core.slot_map, core.used_flags, core.n_slot
            assert self.slot.numel() == n and flat_grads.numel() >= n + self.n_slot
            self.slot_steps = torch.zeros(self.n_slot, dtype=torch.int32, device=dev)
            self.bias_corr = torch.zeros(self.n_slot, 2, dtype=torch.float32, device=dev)
            self.skip_flag = kernel_error_flag(dev)

    def step(self, grad_scale=1.0, zero_grad=True):
        n = self.p.numel()
        self.step_count += 1
        call("grad_sumsq", self.g, self.normsq, n, stream_ptr())
        if self.core is not None:
            call("clip_adamw_step_slots", self.p, self.g, self.m, self.v, n, self.normsq, float(self.max_norm or 0.0),
                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.slot,
                 self.used, self.slot_steps, self.bias_corr, self.n_slot, self.skip_flag, float(grad_scale), int(zero_grad),
                 stream_ptr())
            return
        call("clip_adamw_step", self.p, self.g, self.m, self.v, n, self.normsq, float(self.max_norm or 0.0),
             float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
             self.step_count, float(grad_scale), int(zero_grad), stream_ptr())

    def grad_norm(self):
        """global L2 norm of the gradients seen by the last step (device scalar, no sync)."""
        return self.normsq.sqrt()

    def state_dict(self):
        sd = {"exp_avg": self.m, "exp_avg_sq": self.v, "step": self.step_count, "lr": self.lr}
        if self.core is not None:
            sd["slot_steps"] = self.slot_steps
        return sd

    def load_state_dict(self, sd):
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.lr = float(sd.get("lr", self.lr))
        if self.core is not None:
            if "slot_steps" in sd:
                self.slot_steps.copy_(sd["slot_steps"])
            else:
                self.slot_steps.fill_(self.step_count)
