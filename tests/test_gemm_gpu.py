"""GPU parity: MFMA GEMM kernels vs float64 matmul on the same (bf16-rounded) operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(300, 200, 224), (128, 128, 32), (1000, 3136, 224), (77, 196, 800), (513, 1568, 416),
                                   (2500, 196, 800), (2309, 790, 224), (4100, 1000, 96), (2048, 260, 3136)])
def test_gemm_nt(lib, dtype, M, N, K):
    from urgent2026_challenge_track1_amd import ops
    A, W = _mk((M, K), dtype, 1), _mk((N, K), dtype, 2)
    bias = _mk((N,), torch.float32, 3)
    ref = A.double() @ W.double().T + bias.double()
    got = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), out_dtype=torch.float32).cpu()
    tol = 2e-5 * (K ** 0.5) * 4
    assert (got.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / 10)
    # tanh + bf16 out
    got2 = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), act=1, out_dtype=torch.bfloat16).cpu()
    assert (got2.double() - torch.tanh(ref / 1.0)).abs().max().item() <= 1e-2
    # residual epilogue (in place on an f32 stream) with a strided A view
    res = _mk((M, N), torch.float32, 4)
    Ap = torch.zeros(M, K + 32, dtype=dtype)
    Ap[:, :K] = A
    r = res.clone().cuda()
    ops.gemm_nt(Ap.cuda()[:, :K], W.cuda(), bias.cuda(), resid=r, out=r)
    assert (r.cpu().double() - (ref + res.double())).abs().max().item() <= tol * max(1.0, ref.abs().max().item() / 10)
    if dtype == torch.bfloat16:
        # tanh-backward epilogue: out = (A W^T) * (1 - h^2), h given in the output dtype
        h = torch.tanh(_mk((M, N), torch.float32, 9)).to(torch.bfloat16)
        got3 = ops.gemm_nt(A.cuda(), W.cuda(), None, act=2, resid=h.cuda(), out_dtype=torch.bfloat16).cpu()
        ref3 = (ref - bias.double()) * (1 - h.double() ** 2)
        assert (got3.double() - ref3).abs().max().item() <= 1e-2 * max(1.0, ref3.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(2300, 3136, 224), (2049, 2000, 96), (2177, 500, 160)])
def test_gemm_nt_wide_tile_matches_default(lib, monkeypatch, M, N, K):
    """the 128 x 448 tile (write-bound gate projection) against the 256 x 224 one: ragged M / N edges, every epilogue"""
    from urgent2026_challenge_track1_amd import ops
    A, W = _mk((M, K), torch.bfloat16, 1).cuda(), _mk((N, K), torch.bfloat16, 2).cuda()
    bias = _mk((N,), torch.float32, 3).cuda()
    h = torch.tanh(_mk((M, N), torch.float32, 9)).to(torch.bfloat16).cuda()
    res = _mk((M, N), torch.float32, 4).cuda()
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("URSE_NT_WIDE", mode)
        r = res.clone()
        ops.gemm_nt(A, W, bias, resid=r, out=r)
        outs[mode] = (ops.gemm_nt(A, W, bias, out_dtype=torch.bfloat16), ops.gemm_nt(A, W, bias, out_dtype=torch.float32),
                      ops.gemm_nt(A, W, bias, act=1, out_dtype=torch.bfloat16),
                      ops.gemm_nt(A, W, None, act=2, resid=h, out_dtype=torch.bfloat16), r)
    ref = A.double() @ W.double().T + bias.double()
    assert (outs["1"][1].double() - ref).abs().max().item() <= 2e-5 * (K ** 0.5) * 4 * max(1.0, ref.abs().max().item() / 10)
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)          # same k order per output element -> bit-identical


@pytest.mark.parametrize("M,N,K,act", [(8300, 3136, 224, 0), (9001, 1800, 96, 1), (8200, 2000, 160, 0), (8500, 784, 224, 0), (8193, 500, 128, 1)])
def test_gemm_nt_weight_stationary_matches_ring_kernel(lib, monkeypatch, M, N, K, act):
    """short K, wide N, bf16 out: the weight-stationary kernel (resident 224 x K weight slice, 256-row tiles streamed)
    against the ring kernel - same k order per output element, so bit-identical - and against float64; ragged M and N"""
    from urgent2026_challenge_track1_amd import ops
    A, W = _mk((M, K), torch.bfloat16, 1).cuda(), (_mk((N, K), torch.bfloat16, 2) * 0.2).cuda()
    bias = _mk((N,), torch.float32, 3).cuda()
    ldc = N + 8
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("URSE_NT_BRES", mode)
        C = torch.zeros(M, ldc, dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(A, W, bias, act=act, out=C[:, :N])
        outs[mode] = C
    ref = A.double() @ W.double().T + bias.double()
    if act:
        ref = torch.tanh(ref)
    assert (outs["1"][:, :N].double() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(outs["0"], outs["1"])              # (also: nothing written past column N)


def test_gemm_nt_identity_asymmetric(lib):
    """A = I with an asymmetric B catches a transposed C write or fragment map."""
    from urgent2026_challenge_track1_amd import ops
    K = 64
    A = torch.eye(K, dtype=torch.bfloat16)
    W = (torch.arange(48 * K, dtype=torch.float32).reshape(48, K) % 251 - 100).to(torch.bfloat16)
    got = ops.gemm_nt(A.cuda(), W.cuda(), out_dtype=torch.float32).cpu()
    assert torch.equal(got, W.float().T)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("R,Mo,No", [(1000, 224, 800), (4096, 1568, 416), (333, 3136, 224), (70, 200, 40)])
def test_gemm_tn(lib, dtype, R, Mo, No):
    from urgent2026_challenge_track1_amd import ops
    A, Bm = _mk((R, Mo), dtype, 5), _mk((R, No), dtype, 6)
    ref = A.double().T @ Bm.double()
    out = torch.zeros(Mo - 3, No - 5, device="cuda")
    cs = torch.zeros(Mo - 3, device="cuda")
    ops.gemm_tn(A.cuda(), Bm.cuda(), out, colsum=cs, Mo=Mo - 3, No=No - 5)
    tol = 1e-4 * (R ** 0.5)
    assert (out.cpu().double() - ref[:Mo - 3, :No - 5]).abs().max().item() <= tol
    assert (cs.cpu().double() - A.double().sum(0)[:Mo - 3]).abs().max().item() <= tol
    # accumulate semantics
    ops.gemm_tn(A.cuda(), Bm.cuda(), out, Mo=Mo - 3, No=No - 5)
    assert (out.cpu().double() - 2 * ref[:Mo - 3, :No - 5]).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("inner,period,rev", [(1, 10, False), (1, 10, True), (7, 10, False), (7, 10, True)])
def test_gemm_tn_shifted_operand(lib, inner, period, rev):
    """h_{t-1} operand: B'[r] = B[r -/+ inner], zero at the first/last step of each sequence."""
    from urgent2026_challenge_track1_amd import ops
    R = inner * period * 6
    A, Bm = _mk((R, 64), torch.bfloat16, 7), _mk((R, 96), torch.bfloat16, 8)
    step = (torch.arange(R) // inner) % period
    Bs = torch.zeros_like(Bm)
    if not rev:
        Bs[inner:] = Bm[:-inner]
        Bs[step == 0] = 0
    else:
        Bs[:-inner] = Bm[inner:]
        Bs[step == period - 1] = 0
    ref = A.double().T @ Bs.double()
    out = torch.zeros(64, 96, device="cuda")
    ops.gemm_tn(A.cuda(), Bm.cuda(), out, shift=(inner if rev else -inner), inner=inner, period=period,
                invalid_step=(period - 1 if rev else 0))
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("R,Mo,No,inner,period,rev,perm", [(20000, 608, 200, 1, 0, False, 0), (17 * 34 * 40, 1568, 392, 34, 40, False, 392),
                                                           (17 * 34 * 40, 1568, 392, 34, 40, True, 392), (16500, 3136, 196, 1, 0, False, 392),
                                                           (17000, 196, 784, 1, 0, False, 0), (20000, 200, 600, 1, 0, False, 0),
                                                           (34 * 500 + 7, 1568, 392, 1, 34, False, 392), (34 * 500 + 7, 1568, 392, 1, 34, True, 392),
                                                           (16397, 800, 250, 5, 7, True, 0)])
def test_gemm_tn_large_dma_path(lib, R, Mo, No, inner, period, rev, perm):
    """shapes that take the 256-wide LDS-DMA kernel: ragged tile edges, shifted / masked B rows, gate un-permute,
    column sums, split-R atomics."""
    from urgent2026_challenge_track1_amd import ops
    lda, ldb = (Mo + 7) // 8 * 8 + 8, (No + 31) // 32 * 32
    A, Bm = _mk((R, lda), torch.bfloat16, 11), _mk((R, ldb), torch.bfloat16, 12)
    Bs = Bm.clone()
    kw = {}
    if period:
        step = (torch.arange(R) // inner) % period
        Bs = torch.zeros_like(Bm)
        if not rev:
            Bs[inner:] = Bm[:-inner]
            Bs[step == 0] = 0
        else:
            Bs[:-inner] = Bm[inner:]
            Bs[step == period - 1] = 0
        kw = dict(shift=(inner if rev else -inner), inner=inner, period=period, invalid_step=(period - 1 if rev else 0))
    ref = A.double()[:, :Mo].T @ Bs.double()[:, :No]
    csr = A.double()[:, :Mo].sum(0)
    if perm:   # rows (dir, unit, gate) -> (dir, gate, unit)
        m = torch.arange(Mo)
        d, r = m // (4 * perm), m % (4 * perm)
        dst = d * 4 * perm + (r % 4) * perm + r // 4
        ref2, cs2 = torch.zeros_like(ref), torch.zeros_like(csr)
        ref2[dst], cs2[dst] = ref, csr
        ref, csr = ref2, cs2
    out = torch.zeros(Mo, No, device="cuda")
    cs = torch.zeros(Mo, device="cuda")
    ops.gemm_tn(A.cuda(), Bm.cuda(), out, colsum=cs, Mo=Mo, No=No, perm_h=perm, **kw)
    tol = 1e-4 * (R ** 0.5)
    assert (out.cpu().double() - ref).abs().max().item() <= tol
    assert (cs.cpu().double() - csr).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_grouped_matches_single_calls(lib, dtype):
    from urgent2026_challenge_track1_amd import ops
    shapes = [(900, 64, 40), (900, 12, 784), (1300, 784, 200), (77, 200, 8)]
    rows, outs, refs, css, keep = [], [], [], [], []
    for i, (R, Mo, No) in enumerate(shapes):
        A, Bm = _mk((R, (Mo + 15) // 8 * 8), dtype, 20 + i).cuda(), _mk((R, (No + 15) // 8 * 8), dtype, 40 + i).cuda()
        keep += [A, Bm]                                   # descriptors hold raw pointers
        out, cs = torch.zeros(Mo, No, device="cuda"), torch.zeros(Mo, device="cuda")
        rows.append(ops.tn_desc(A, Bm, out, colsum=cs, Mo=Mo, No=No))
        ref, rcs = torch.zeros(Mo, No, device="cuda"), torch.zeros(Mo, device="cuda")
        ops.gemm_tn(A, Bm, ref, colsum=rcs, Mo=Mo, No=No)
        outs.append(out); refs.append(ref); css.append((cs, rcs))
    ops.gemm_tn_grouped(rows, dtype, "cuda")
    for out, ref, (cs, rcs) in zip(outs, refs, css):
        tol = 1e-5 * max(1.0, ref.abs().max().item())        # same products, different split-R summation order
        assert (out - ref).abs().max().item() <= tol
        assert (cs - rcs).abs().max().item() <= 1e-5 * max(1.0, rcs.abs().max().item())


@pytest.mark.parametrize("dtype,R,H,N,inner,period,rev", [(torch.bfloat16, 34 * 20 * 25, 392, 196, 34, 25, False),
                                                          (torch.bfloat16, 34 * 20 * 25, 392, 196, 34, 25, True),
                                                          # whole 32-row stages and whole (inner x period) blocks: the 224 x 320 tile kernel
                                                          (torch.bfloat16, 34 * 32 * 16, 392, 196, 34, 32, False),
                                                          (torch.bfloat16, 34 * 32 * 16, 392, 196, 34, 32, True),
                                                          (torch.bfloat16, 34 * 544, 392, 196, 1, 34, False),
                                                          (torch.bfloat16, 34 * 544, 392, 196, 1, 34, True),
                                                          (torch.float32, 600, 24, 16, 5, 12, False)])
def test_gemm_tn_dual_matches_two_calls(lib, dtype, R, H, N, inner, period, rev):
    """both weight gradients of one LSTM direction in one pass over the dgates == the two separate contractions
    (large bf16 shapes take the dual-operand ring kernel, everything else falls back to two calls)."""
    from urgent2026_challenge_track1_amd import ops
    A = _mk((R, 4 * H), dtype, 30).cuda()
    X = _mk((R, (N + 31) // 32 * 32), dtype, 31).cuda()
    Hh = _mk((R, (H + 31) // 32 * 32), dtype, 32).cuda()
    sh, inv = (inner, period - 1) if rev else (-inner, 0)
    c1, c2, cs = torch.zeros(4 * H, N, device="cuda"), torch.zeros(4 * H, H, device="cuda"), torch.zeros(4 * H, device="cuda")
    r1, r2, rs = torch.zeros_like(c1), torch.zeros_like(c2), torch.zeros_like(cs)
    ops.gemm_tn_dual(A, X, c1, cs, Hh, c2, 4 * H, N, H, sh, inner, period, inv, perm_h=H)
    ops.gemm_tn(A, X, r1, colsum=rs, Mo=4 * H, No=N, perm_h=H)
    ops.gemm_tn(A, Hh, r2, Mo=4 * H, No=H, shift=sh, inner=inner, period=period, invalid_step=inv, perm_h=H)
    for got, ref in ((c1, r1), (c2, r2), (cs, rs)):
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("act,out_dtype", [(0, torch.float32), (1, torch.bfloat16), (2, torch.bfloat16)])
def test_gemm_nt_grouped_h_ring_kernel_matches_small_kernel(lib, monkeypatch, act, out_dtype):
    """grouped records (one per band: own N, K, pointers) on the LDS-DMA ring kernel against the 128 x 128 kernel and
    against float64, every epilogue; M not a multiple of the tile"""
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd.bsrnn import nt_grouped
    M, shapes = 1300, [(784, 224), (196, 800), (160, 32), (230, 96)]
    keep, rows, refs = [], [], []
    for g, (N, K) in enumerate(shapes):
        A, W = _mk((M, K), torch.bfloat16, 10 + g).cuda(), (_mk((N, K), torch.bfloat16, 20 + g) * 0.2).cuda()
        bias = _mk((N,), torch.float32, 30 + g).cuda() if act != 2 else None
        h = torch.tanh(_mk((M, N), torch.float32, 40 + g)).to(out_dtype).cuda() if act == 2 else None
        ldc = N + 8
        C = torch.zeros(M, ldc, dtype=out_dtype, device="cuda")
        keep += [A, W, bias, h, C]
        rows.append([A.data_ptr(), W.data_ptr(), C.data_ptr(), 0 if bias is None else bias.data_ptr(),
                     0 if h is None else h.data_ptr(), K, K, ldc, M, N, K, 0 if h is None else N])
        ref = A.double() @ W.double().T
        if act == 1:
            ref = torch.tanh(ref + bias.double())
        elif act == 2:
            ref = ref * (1 - h.double() ** 2)
        else:
            ref = ref + bias.double()
        refs.append((C, N, ref))
    outs = {}
    for mode in ("ring", "small"):
        if mode == "small":
            monkeypatch.setenv("URSE_NT_GROUPED_NO_DMA", "1")
        for C, _, _ in refs:
            C.zero_()
        nt_grouped(rows, "cuda", ops.BF16, ops._dt(refs[0][0]), act=act)
        outs[mode] = [C.clone() for C, _, _ in refs]
    for (C, N, ref), a, b in zip(refs, outs["ring"], outs["small"]):
        tol = 1e-2 if out_dtype == torch.bfloat16 else 2e-3
        assert (a[:, :N].double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        assert (a[:, :N].double() - b[:, :N].double()).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
        assert torch.count_nonzero(a[:, N:]) == 0          # nothing written past a group's N


@pytest.mark.parametrize("M,N,K,rows,resid", [(4096, 196, 800, 1024, True), (4096, 196, 800, 2048, False), (3 * 640, 196, 416, 640, True),
                                              (512, 20, 64, 128, True)])
def test_gemm_nt_with_groupnorm_statistics(lib, M, N, K, rows, resid):
    """urse_gemm_nt_gnstats: the output equals urse_gemm_nt's bit for bit, and the (sum, sum of squares) per group of rows equal
    the statistics pass on that output - from the ring kernel's epilogue (first three shapes) and from the fallback (last)."""
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g) if resid else None
    ref = ops.gemm_nt(A, W, bias, resid=r, out_dtype=torch.float32)
    out, st = ops.gemm_nt(A, W, bias, resid=r, out_dtype=torch.float32, gn_rows=rows)
    assert torch.equal(out, ref)
    x = ref.double().view(M // rows, rows * N)
    want = torch.stack([x.sum(1), (x * x).sum(1)], 1).reshape(-1)
    assert (st - want).abs().max().item() <= 1e-6 * want.abs().max().item(), (st, want)
    # and the normalisation that takes them equals the two-pass one
    gamma, beta = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    Np = (N + 31) // 32 * 32
    y1, s1 = ops.groupnorm_fwd(ref, gamma, beta, M // rows, rows, 1, N, N, Np, 0, torch.bfloat16)
    y2, _ = ops.groupnorm_fwd(ref, gamma, beta, M // rows, rows, 1, N, N, Np, 0, torch.bfloat16, stats=st)
    assert (y1.float() - y2.float()).abs().max().item() <= 1e-2 and (s1 - st).abs().max().item() <= 1e-6 * want.abs().max().item()


@pytest.mark.parametrize("M,N,K,rows", [(4096, 196, 3136, 1024), (6 * 640, 196, 416, 640), (5 * 512 + 0, 196, 800, 512)])
def test_dgrad_gemm_with_groupnorm_backward_sums(lib, M, N, K, rows):
    """urse_gemm_nt_gnbwd + urse_groupnorm_bwd_apply == urse_gemm_nt + urse_groupnorm_bwd: the dgrad output bit for bit, the group sums,
    dgamma / dbeta and dx to f32 rounding (different summation order).  Tiles that span two groups (rows not a multiple of 256) included."""
    from urgent2026_challenge_track1_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    Bn = M // rows
    x = torch.randn(Bn, rows, 1, N, device="cuda", generator=g) * 2 + 0.3
    gamma, beta = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    dres = torch.randn(Bn, rows, 1, N, device="cuda", generator=g)
    Np = (N + 31) // 32 * 32
    _, stats = ops.groupnorm_fwd(x, gamma, beta, Bn, rows, 1, N, N, Np, 0, torch.bfloat16)
    assert ops.gemm_nt_gnbwd_supported(M, N, K, rows, torch.bfloat16)
    # two-pass reference
    dy0 = ops.gemm_nt(A, W, out_dtype=torch.float32)
    dg0, db0 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    dx0, p0 = ops.groupnorm_bwd(x, dy0, stats, gamma, dres, dg0, db0, Bn, rows, 1, N, N, 0, pack_ld=Np)
    # fused
    dg1, db1 = torch.ones(N, device="cuda"), torch.ones(N, device="cuda")          # (accumulated: += on top of what is there)
    dy1, sums = ops.gemm_nt_gnbwd(A, W, N, x, stats, gamma, dg1, db1, rows)
    dx1, p1 = ops.groupnorm_bwd(x, dy1, stats, gamma, dres, dg1, db1, Bn, rows, 1, N, N, 0, pack_ld=Np, sums=sums)
    assert torch.equal(dy0, dy1)
    xd = x.double().view(Bn, rows * N)
    mu = xd.mean(1, keepdim=True)
    xh = ((xd - mu) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)).view(M, N)
    dyd = dy0.double()
    want = torch.stack([(dyd * gamma.double()).view(Bn, -1).sum(1), (dyd * gamma.double() * xh).view(Bn, -1).sum(1)], 1).reshape(-1)
    scale = (dyd.abs() * gamma.double()).view(Bn, -1).sum(1).max().item()
    assert (sums - want).abs().max().item() <= 1e-5 * scale, (sums, want)
    for a, b in ((dg1 - 1, dg0), (db1 - 1, db0)):
        assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item()), (a - b).abs().max()
    assert (dx1 - dx0).abs().max().item() <= 1e-4 * max(1.0, dx0.abs().max().item())
    assert (p1.float() - p0.float()).abs().max().item() <= 2e-2 * max(1.0, dx0.abs().max().item())


def _poisoned_slice(rows, cols, ld_extra, row_extra, dtype, seed, tail=True):
    """an operand that is the LAST `rows` rows and the LAST `cols` columns of its allocation: the parent is exactly
    (row_extra + rows) x (ld_extra + cols) elements, every byte outside the slice is NaN.  Returns (slice view on the GPU, clean CPU copy)."""
    clean = _mk((rows, cols), dtype, seed)
    parent = torch.full((row_extra + rows, ld_extra + cols), float("nan"), dtype=dtype)
    parent[row_extra:, ld_extra:] = clean
    dev = parent.cuda()
    return dev[row_extra:, ld_extra:], clean, dev


@pytest.mark.parametrize("kind", ["tn_128", "tn_ring", "tn_ring_perm", "tn_dual224", "tn_grouped", "nt_ring", "nt_bres", "nt_128"])
def test_gemm_operand_slices_at_the_end_of_their_allocation_with_poison_behind(lib, kind):
    """VERDICT r4 (the 11fd85f class of bug: a column-slice operand read past its parent's last row): every GEMM family with operands that
    END their allocation (last rows, last columns) and NaN in every byte of the parent outside the slice - a read that strays outside the
    slice and reaches the arithmetic makes the result NaN; results must equal the call on clean contiguous copies."""
    from urgent2026_challenge_track1_amd import ops
    bf = torch.bfloat16
    ops.launch_counts(reset=True)
    if kind.startswith("tn"):
        if kind == "tn_128":
            R, Mo, No, kw, dt = 333, 200, 40, {}, torch.float32
        elif kind == "tn_ring":
            R, Mo, No, kw, dt = 20000, 608, 200, {}, bf
        elif kind == "tn_ring_perm":
            R, Mo, No, kw, dt = 34 * 500 + 7, 1568, 392, dict(shift=-1, inner=1, period=34, invalid_step=0, perm_h=392), bf
        elif kind == "tn_dual224":
            R, H, N, inner, period = 34 * 32 * 16, 392, 196, 34, 32
            A, Ac, keepA = _poisoned_slice(R, 4 * H, 4 * H, 3, bf, 30)          # direction 1's half of an [M, 8H] dgates matrix
            X, Xc, keepX = _poisoned_slice(R, 224, 32, 5, bf, 31)
            Hh, Hc, keepH = _poisoned_slice(R, 416, 416, 2, bf, 32)               # direction 1's half of hout
            Xc[:, N:] = 0; Hc[:, H:] = 0
            X[:, N:] = 0; Hh[:, H:] = 0                                         # K padding the kernels may read must be zero (it is in the model)
            c1, c2, cs = (torch.zeros(4 * H, N, device="cuda"), torch.zeros(4 * H, H, device="cuda"), torch.zeros(4 * H, device="cuda"))
            ops.gemm_tn_dual(A, X, c1, cs, Hh, c2, 4 * H, N, H, -inner, inner, period, 0, perm_h=H)
            r1, r2, rs = torch.zeros_like(c1), torch.zeros_like(c2), torch.zeros_like(cs)
            ops.gemm_tn_dual(Ac.cuda(), Xc.cuda(), r1, rs, Hc.cuda(), r2, 4 * H, N, H, -inner, inner, period, 0, perm_h=H)
            assert ops.launch_counts()["tn_dual"] == 2
            for got, ref in ((c1, r1), (c2, r2), (cs, rs)):
                assert torch.isfinite(got).all(), kind
                assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
            return
        elif kind == "tn_grouped":
            rows, outs, refs, keep = [], [], [], []
            for i, (R, Mo, No) in enumerate([(900, 64, 40), (900, 12, 784), (1300, 784, 200), (77, 200, 8)]):
                A, Ac, k1 = _poisoned_slice(R, (Mo + 7) // 8 * 8, 16, 1, bf, 20 + i)
                Bm, Bc, k2 = _poisoned_slice(R, (No + 7) // 8 * 8, 24, 1, bf, 40 + i)
                out = torch.zeros(Mo, No, device="cuda")
                rows.append(ops.tn_desc(A, Bm, out, Mo=Mo, No=No))
                keep += [k1, k2]
                outs.append(out)
                refs.append(Ac.double()[:, :Mo].T @ Bc.double()[:, :No])
            ops.gemm_tn_grouped(rows, bf, "cuda")
            for out, ref in zip(outs, refs):
                assert torch.isfinite(out).all()
                assert (out.cpu().double() - ref).abs().max().item() <= 1e-4 * (1300 ** 0.5)
            return
        lda = (Mo + 7) // 8 * 8
        ldb = (No + 31) // 32 * 32
        A, Ac, k1 = _poisoned_slice(R, lda, 40, 2, dt, 11)
        Bm, Bc, k2 = _poisoned_slice(R, ldb, 64, 2, dt, 12)
        out, cs = torch.zeros(Mo, No, device="cuda"), torch.zeros(Mo, device="cuda")
        ref, rcs = torch.zeros_like(out), torch.zeros_like(cs)
        ops.gemm_tn(A, Bm, out, colsum=cs, Mo=Mo, No=No, **kw)
        ops.gemm_tn(Ac.cuda(), Bc.cuda(), ref, colsum=rcs, Mo=Mo, No=No, **kw)
        counts = ops.launch_counts()
        assert counts["tn_128" if kind == "tn_128" else ("tn_ring_t" if counts["tn_ring_t"] else "tn_ring")] == 2, counts
        assert torch.isfinite(out).all() and torch.isfinite(cs).all(), kind
        tol = 2e-5 * max(1.0, ref.abs().max().item())            # split-R atomics: summation order differs between two calls
        assert (out - ref).abs().max().item() <= tol and (cs - rcs).abs().max().item() <= 2e-5 * max(1.0, rcs.abs().max().item())
        return
    M, N, K, dt = {"nt_ring": (20000, 196, 800, bf), "nt_bres": (16500, 3136, 224, bf), "nt_128": (300, 200, 224, torch.float32)}[kind]
    A, Ac, k1 = _poisoned_slice(M, K, 32, 3, dt, 1)
    W, Wc, k2 = _poisoned_slice(N, K, 64, 2, dt, 2)
    bias = _mk((N,), torch.float32, 3).cuda()
    odt = torch.bfloat16 if kind == "nt_bres" else torch.float32      # (the weight-stationary kernel is the bf16-out gate projection)
    got = ops.gemm_nt(A, W, bias, out_dtype=odt)
    ref = ops.gemm_nt(Ac.cuda(), Wc.cuda(), bias, out_dtype=odt)
    counts = ops.launch_counts()
    assert counts[kind] == 2, counts
    assert torch.isfinite(got.float()).all(), kind
    assert torch.equal(got, ref)          # same k order per element, no atomics: bit-equal
    # the output too: a slice that ends its allocation, poison in front; nothing outside the slice may be written
    outp = torch.full((M + 2, N + 8), float("nan"), device="cuda", dtype=odt)
    ops.gemm_nt(A, W, bias, out=outp[2:, 8:])
    assert torch.equal(outp[2:, 8:], ref) and torch.isnan(outp[:2].float()).all() and torch.isnan(outp[:, :8].float()).all()


@pytest.mark.parametrize("inner,period,rev", [(34, 25, False), (34, 25, True), (1, 34, False), (1, 34, True)])
def test_gemm_tn_dual_bf16_gradients_against_f16_activations(lib, inner, period, rev):
    """URSE_BF16_ACT_F16 (round 6; the nn.LSTM weight gradients of `d_model.py:61-89`'s backward in an f16-forward step): A = bf16 gate
    gradients, B / B2 = the forward's IEEE-half x_n / h, converted to bf16 in registers behind the LDS fragment read.  Exactly the product of
    A with the bf16-ROUNDED activations: equal to the all-bf16 kernel fed `B.float().bfloat16()` up to the order of the f32 atomics, and
    within bf16 rounding of the float64 product with the unrounded f16 values."""
    from urgent2026_challenge_track1_amd import ops
    H, N = 392, 196
    R = inner * period * (32 if inner > 1 else 640)
    assert R % 32 == 0 and ops.tn_act_f16_supported(R, 4 * H, N, H, True, inner, period)
    A = (_mk((R, 4 * H), torch.float32, 50) * 0.3).bfloat16().cuda()
    X = torch.zeros(R, 224, dtype=torch.float16)
    X[:, :N] = _mk((R, N), torch.float32, 51).half()
    Hh = torch.zeros(R, 416, dtype=torch.float16)
    Hh[:, :H] = torch.tanh(_mk((R, H), torch.float32, 52)).half()
    X, Hh = X.cuda(), Hh.cuda()
    sh, inv = (inner, period - 1) if rev else (-inner, 0)
    c1, c2, cs = torch.zeros(4 * H, N, device="cuda"), torch.zeros(4 * H, H, device="cuda"), torch.zeros(4 * H, device="cuda")
    r1, r2, rs = torch.zeros_like(c1), torch.zeros_like(c2), torch.zeros_like(cs)
    ops.launch_counts(reset=True)
    ops.gemm_tn_dual(A, X, c1, cs, Hh, c2, 4 * H, N, H, sh, inner, period, inv, perm_h=H)
    assert ops.launch_counts()["tn_dual"] == 1
    ops.gemm_tn_dual(A, X.float().bfloat16(), r1, rs, Hh.float().bfloat16(), r2, 4 * H, N, H, sh, inner, period, inv, perm_h=H)
    for got, ref in ((c1, r1), (c2, r2), (cs, rs)):
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # float64 of the same contraction with the f16 values themselves (un-permuted rows: perm_h maps (unit, gate) -> (gate, unit))
    c1p = torch.zeros(4 * H, N, device="cuda")
    ops.gemm_tn_dual(A, X, c1p, None, Hh, torch.zeros_like(c2), 4 * H, N, H, sh, inner, period, inv, perm_h=0)
    ref = (A.double().T @ X.double()[:, :N])
    assert ((c1p.double() - ref).norm() / ref.norm()).item() < 3e-3


def test_gemm_tn_fc_gradient_bf16_against_f16_h(lib):
    """the wide-and-short fc weight gradient [196, 784] (transposed ring kernel, column sums of the gradient operand) with h in IEEE half."""
    from urgent2026_challenge_track1_amd import ops
    R, N, H2 = 34 * 640, 196, 784
    assert ops.tn_act_f16_supported(R, N, H2, 0, True)
    dO = torch.zeros(R, 224, dtype=torch.bfloat16)
    dO[:, :N] = (_mk((R, N), torch.float32, 60) * 0.2).bfloat16()
    Hh = torch.zeros(R, 800, dtype=torch.float16)
    Hh[:, :H2] = torch.tanh(_mk((R, H2), torch.float32, 61)).half()
    dO, Hh = dO.cuda(), Hh.cuda()
    c, cs = torch.zeros(N, H2, device="cuda"), torch.zeros(N, device="cuda")
    r, rs = torch.zeros_like(c), torch.zeros_like(cs)
    ops.launch_counts(reset=True)
    ops.gemm_tn(dO, Hh, c, colsum=cs, Mo=N, No=H2)
    assert ops.launch_counts()["tn_ring_t"] == 1
    ops.gemm_tn(dO, Hh.float().bfloat16(), r, colsum=rs, Mo=N, No=H2)
    for got, ref in ((c, r), (cs, rs)):
        assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    ref = dO.double()[:, :N].T @ Hh.double()[:, :H2]
    assert ((c.double() - ref).norm() / ref.norm()).item() < 3e-3


def test_gemm_tn_mixed_operands_refuse_shapes_without_a_kernel(lib):
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd._lib import UrseError
    A = torch.zeros(64, 224, dtype=torch.bfloat16, device="cuda")
    Bm = torch.zeros(64, 64, dtype=torch.float16, device="cuda")
    assert not ops.tn_act_f16_supported(64, 224, 64, 0, True)
    with pytest.raises(UrseError):
        ops.gemm_tn(A, Bm, torch.zeros(224, 64, device="cuda"), Mo=224, No=64)
