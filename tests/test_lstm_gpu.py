"""GPU parity: LSTM recurrence kernels (fwd + BPTT) vs torch.nn.LSTM on CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, T, K, N, path, dtype, seed=0):
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(seed)
    H = 2 * N
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    x = torch.randn(B, T, K, N)                      # channel-last activations
    if path == "time":
        seqs = x.permute(0, 2, 1, 3).reshape(B * K, T, N)
        n_seq, seq_len, inner, outer, stride = B * K, T, K, T * K, K
    else:
        seqs = x.reshape(B * T, K, N)
        n_seq, seq_len, inner, outer, stride = B * T, K, 1, K, 1
    seqs = seqs.clone().requires_grad_(True)
    y, _ = lstm(seqs)
    gy = torch.randn_like(y)
    y.backward(gy)
    if path == "time":
        y_rows = y.detach().reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
        gy_rows = gy.reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
        gx_rows = seqs.grad.reshape(B, K, T, N).permute(0, 2, 1, 3).reshape(-1, N)
    else:
        y_rows, gy_rows, gx_rows = y.detach().reshape(-1, 2 * H), gy.reshape(-1, 2 * H), seqs.grad.reshape(-1, N)

    dev = "cuda"
    M = B * T * K
    Np, Hp = ops.kpad(N, dtype), ops.kpad(ops.pad_to(H, 16), dtype)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, Np, dtype)
    wih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse]).detach().to(dev).contiguous()
    whh = torch.cat([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse]).detach().to(dev).contiguous()
    bih = torch.cat([lstm.bias_ih_l0, lstm.bias_ih_l0_reverse]).detach().to(dev).contiguous()
    bhh = torch.cat([lstm.bias_hh_l0, lstm.bias_hh_l0_reverse]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(wih, whh, bih, bhh, N, H, dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, Hp, n_seq, seq_len, inner, outer, stride)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    err = (hout[:, :2 * H].float().cpu() - y_rows).abs().max().item()
    assert err <= tol, ("fwd", err)
    assert torch.all(hout[:, 2 * H:] == 0)

    # backward through time
    ldh = hout.shape[1]
    dh = ops.pack2d(gy_rows.to(dev), M, ldh, dtype)
    dg = ops.lstm_bwd(dh, gx, c, pk["whhT"], H, n_seq, seq_len, inner, outer, stride)
    # dx = dgates @ W_ih
    dx = ops.gemm_nt(dg, pk["wihT"], out_dtype=torch.float32)
    scale = gx_rows.abs().max().item()
    err = (dx.cpu() - gx_rows).abs().max().item()
    assert err <= (3e-2 if dtype == torch.bfloat16 else 5e-5) * scale, ("dx", err, scale)
    # weight grads
    dwih = torch.zeros(8 * H, N, device=dev)
    db = torch.zeros(8 * H, device=dev)
    ops.gemm_tn(dg, xr, dwih, colsum=db, No=N, perm_h=H)
    ref_dwih = torch.cat([lstm.weight_ih_l0.grad, lstm.weight_ih_l0_reverse.grad])
    ref_db = torch.cat([lstm.bias_ih_l0.grad, lstm.bias_ih_l0_reverse.grad])
    wt = (3e-2 if dtype == torch.bfloat16 else 1e-4)
    assert (dwih.cpu() - ref_dwih).abs().max().item() <= wt * ref_dwih.abs().max().item()
    assert (db.cpu() - ref_db).abs().max().item() <= wt * ref_db.abs().max().item()
    for d, (wname, inv, sh) in enumerate([("weight_hh_l0", 0, -stride), ("weight_hh_l0_reverse", seq_len - 1, stride)]):
        dwhh = torch.zeros(4 * H, H, device=dev)
        ops.gemm_tn(dg[:, d * 4 * H:(d + 1) * 4 * H], hout[:, d * H:(d + 1) * H], dwhh, shift=sh, inner=stride,
                    period=seq_len, invalid_step=inv, perm_h=H)
        ref = getattr(lstm, wname).grad
        assert (dwhh.cpu() - ref).abs().max().item() <= wt * ref.abs().max().item(), wname


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("path", ["time", "band"])
@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 16), (1, 33, 5, 24), (3, 7, 34, 196)])
def test_lstm_fwd_bwd(lib, dtype, path, B, T, K, N):
    _case(B, T, K, N, path, dtype)


@pytest.mark.parametrize("path", ["time", "band"])
@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 16), (3, 7, 34, 196), (2, 40, 34, 196)])
def test_cluster_kernel_matches_streaming_kernel(lib, path, B, T, K, N):
    """persistent cluster LSTM (weights in registers, cross-CU h exchange) == streaming kernel, bit for bit on h/c
    (same bf16 MFMA products, same f32 cell math)."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(1)
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    assert ops.lstm_cluster_plan(H, pk["Hp"], sm["n_seq"]) is not None
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx1 = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    gx2 = gx1.clone()
    h1, c1 = ops.lstm_fwd(gx1, pk["whh"], H, pk["Hp"], **sm)
    h2, c2, err = ops.lstm_fwd_cluster(gx2, pk["whhq"], H, pk["Hp"], **sm)
    assert int(err.item()) == 0
    assert (h1.float() - h2.float()).abs().max().item() <= 1e-2      # a bf16 ulp where accumulation order differs
    assert (c1 - c2).abs().max().item() <= 2e-2
    assert (gx1.float() - gx2.float()).abs().max().item() <= 2e-2   # saved gate activations
    assert (h1.float() - h2.float()).abs().mean().item() <= 1e-4


@pytest.mark.parametrize("path", ["time", "band"])
@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 16), (3, 7, 34, 196), (2, 40, 34, 196)])
def test_cluster_bptt_matches_streaming_kernel(lib, path, B, T, K, N):
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(2)
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, hout.shape[1], dtype)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, **sm)
    _, err = ops.lstm_bwd_cluster(dh, g2, c, pk["whhTq"], H, pk["Hp"], **sm)
    assert int(err.item()) == 0
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-4 * scale, (d.max().item(), d.mean().item(), scale)


@pytest.mark.parametrize("path", ["time", "band"])
@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 16), (1, 33, 5, 24), (3, 7, 34, 196), (5, 40, 34, 196)])
def test_wide_kernel_matches_streaming_kernel(lib, path, B, T, K, N):
    """wide LSTM forward (64 sequences per workgroup, block-ordered weights) == streaming kernel up to the f32
    accumulation order (the pre-activation enters the accumulator first instead of last)."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(3)
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    assert "whhb" in pk
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx1 = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    gx2 = gx1.clone()
    h1, c1 = ops.lstm_fwd(gx1, pk["whh"], H, pk["Hp"], **sm)
    h2, c2 = ops.lstm_fwd_wide(gx2, pk["whhb"], H, pk["Hp"], **sm)
    assert torch.all(h2[:, 2 * H:] == 0)
    assert (h1.float() - h2.float()).abs().max().item() <= 1e-2
    assert (c1 - c2).abs().max().item() <= 2e-2
    assert (gx1.float() - gx2.float()).abs().max().item() <= 2e-2
    assert (h1.float() - h2.float()).abs().mean().item() <= 1e-4


@pytest.mark.parametrize("path", ["time", "band"])
def test_bptt_32_row_variant_matches_16_row_variant(lib, path):
    """rows16 = 18 (32 sequences / 8 waves per workgroup, picked automatically for the C2 band path) vs rows16 = 1."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(4)
    B, T, K, N = 3, 7, 34, 196
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, hout.shape[1], dtype)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, rows16=1, **sm)
    ops.lstm_bwd(dh, g2, c, pk["whhT"], H, rows16=18, **sm)
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-4 * scale, (d.max().item(), d.mean().item(), scale)


@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 48), (3, 7, 34, 196), (2, 40, 34, 196), (5, 25, 33, 196), (2, 21, 48, 384), (1, 12, 13, 384)])
def test_split_bptt_matches_streaming_kernel(lib, B, T, K, N):
    """time-path BPTT split over 2-3 workgroups per 32 sequences (f32 partial sums exchanged with tagged data) vs the
    one-workgroup streaming kernel: same bf16 products, partial sums lose their LSB and are added in a different order."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(5)
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    assert ops.lstm_split_plan(H, sm["n_seq"]) is not None
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, hout.shape[1], dtype)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, **sm)
    _, err = ops.lstm_bwd_split(dh, g2, c, pk["whhT"], H, **sm)
    assert int(err.item()) == 0
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-4 * scale, (d.max().item(), d.mean().item(), scale)


def test_split_bptt_chunked_band_path_matches_streaming_kernel(lib, monkeypatch):
    """band path of the flow model (H = 768): 1,002 sequences of 12 steps exceed what one split launch can keep resident; the BPTT
    runs as row-block launches (opt-in: URSE_LSTM_SPLIT_BWD_MAX_CHUNKS) == the streaming kernel."""
    from urgent2026_challenge_track1_amd import ops
    monkeypatch.setattr(ops, "SPLIT_BWD_MAX_CHUNKS", 4)
    torch.manual_seed(12)
    N, B, T, K = 384, 2, 501, 12
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    assert ops.lstm_split_plan(H, sm["n_seq"]) is None
    chunks = ops.lstm_split_chunks(H, **sm)
    assert chunks is not None and len(chunks) >= 2 and sum(n for _, n in chunks) == 1002
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, hout.shape[1], dtype)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, **sm)
    ops.launch_counts(reset=True)
    _, err = ops.lstm_bwd_split(dh, g2, c, pk["whhT"], H, **sm)
    assert int(err.item()) == 0 and ops.launch_counts()["lstm_bwd_split"] == len(chunks)
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-4 * scale, (d.max().item(), d.mean().item(), scale)


@pytest.mark.parametrize("path,B,T,K,N", [("time", 2, 21, 48, 384), ("band", 1, 60, 10, 384), ("time", 3, 40, 34, 196), ("time", 1, 9, 5, 196)])
def test_cluster2_kernel_matches_streaming_kernel(lib, path, B, T, K, N):
    """generalised cluster forward (H = 768: 24 workgroups per cluster; H = 392: two unit quads per wave) == streaming kernel."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(6)
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    assert ops.lstm_cluster2_plan(H, pk["Hp"], sm["n_seq"]) is not None
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx1 = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    gx2 = gx1.clone()
    h1, c1 = ops.lstm_fwd(gx1, pk["whh"], H, pk["Hp"], **sm)
    h2, c2, err = ops.lstm_fwd_cluster2(gx2, pk["whhq"], H, pk["Hp"], **sm)
    assert int(err.item()) == 0
    assert torch.all(h2[:, 2 * H:] == 0)
    assert (h1.float() - h2.float()).abs().max().item() <= 1e-2
    assert (c1 - c2).abs().max().item() <= 2e-2
    assert (gx1.float() - gx2.float()).abs().max().item() <= 2e-2
    assert (h1.float() - h2.float()).abs().mean().item() <= 1e-4


def test_cluster2_chunked_band_path_matches_streaming_kernel(lib):
    """more sequences than the resident clusters hold (H = 768: 320 per direction) run as several launches over contiguous row
    blocks of the band path: B x T = 700 sequences of K = 12 steps == the streaming kernel, saved activations and cell state too."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(8)
    N, B, T, K = 384, 1, 700, 12
    H, dtype, dev = 2 * N, torch.bfloat16, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    M = B * T * K
    sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    assert ops.lstm_cluster2_plan(H, pk["Hp"], sm["n_seq"]) is None
    chunks = ops.lstm_cluster2_chunks(H, pk["Hp"], **sm)
    assert chunks is not None and len(chunks) == 3 and sum(n for _, n in chunks) == 700
    assert ops.lstm_cluster2_chunks(H, pk["Hp"], n_seq=700, seq_len=K, inner=K, outer=T * K, stride=K) is None   # not row blocks
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    gx1 = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    gx2 = gx1.clone()
    h1, c1 = ops.lstm_fwd(gx1, pk["whh"], H, pk["Hp"], **sm)
    ops.launch_counts(reset=True)
    h2, c2, err = ops.lstm_fwd_cluster2(gx2, pk["whhq"], H, pk["Hp"], **sm)
    assert int(err.item()) == 0 and ops.launch_counts()["lstm_fwd_cluster2"] == 3
    assert (h1.float() - h2.float()).abs().max().item() <= 1e-2
    assert (c1 - c2).abs().max().item() <= 2e-2
    assert (gx1.float() - gx2.float()).abs().max().item() <= 2e-2
    assert (h1.float() - h2.float()).abs().mean().item() <= 1e-4


def test_cooperative_kernel_timeout_is_reported(lib):
    """the cluster / split kernels share one device error word; a set word makes the next check raise (fail loudly)."""
    from urgent2026_challenge_track1_amd import ops, _lib
    flag = ops.kernel_error_flag(torch.device("cuda", torch.cuda.current_device()))
    dev = flag.device
    ops.poll_kernel_errors(dev, sync=True)                      # clean
    flag.fill_(1)
    with pytest.raises(_lib.UrseError):
        ops.poll_kernel_errors(dev, sync=True)
    ops.poll_kernel_errors(dev)                                  # deferred form: first call only requests the copy ...
    torch.cuda.synchronize()
    with pytest.raises(_lib.UrseError):
        ops.poll_kernel_errors(dev)                              # ... the next one sees it
    flag.zero_()
    torch.cuda.synchronize()
    try:
        ops.poll_kernel_errors(dev)                              # reads the copy requested while the word was still set
    except _lib.UrseError:
        pass
    torch.cuda.synchronize()
    ops.poll_kernel_errors(dev)                                  # clean again


@pytest.mark.parametrize("sm", [dict(n_seq=100, seq_len=5, inner=1, outer=5, stride=1),          # ragged last tile, 7 workgroups of one tile
                                dict(n_seq=1, seq_len=3, inner=1, outer=3, stride=1),
                                dict(n_seq=2 * 34, seq_len=9, inner=34, outer=9 * 34, stride=34),   # time-path row map
                                dict(n_seq=1900, seq_len=34, inner=1, outer=34, stride=1)])       # 119 tiles: workgroups of 1 and 0 ... 7 tiles
@pytest.mark.parametrize("save", [True, False])
def test_row_wave_forward_equals_wide_forward(lib, sm, save):
    """csrc/lstm_rw.hip (16 sequences per wave, W_hh shared through an LDS-DMA ring) == csrc/lstm_wide.hip bit for bit: h, c and the
    saved gate activations (same bf16 MFMA products in the same order, same f32 cell math).  (The paired form - two adjacent units per lane,
    round 4, bit-identical too and no faster - is compiled into variant builds only since round 6; the shipped library refuses it.)"""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(3)
    N, dev = 196, "cuda"
    H, Hp = 2 * N, 416
    assert ops.lstm_rw_supported(H, Hp)
    whh = torch.randn(2 * 4 * H, H, device=dev) * 0.05
    whhb = torch.empty(2 * 25 * 13 * 4 * 512, device=dev, dtype=torch.bfloat16)
    ops.call("lstm_pack_blocks", whh, whhb, H, Hp, ops.stream_ptr())
    whhb_rw = torch.empty_like(whhb)
    ops.call("lstm_pack_blocks_rw", whh, whhb_rw, H, Hp, ops.stream_ptr())
    M = sm["n_seq"] * sm["seq_len"]
    gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
    g1, g2 = gx.clone(), gx.clone()
    h1, c1 = ops.lstm_fwd_wide(g1, whhb, H, Hp, save=save, **sm)
    from urgent2026_challenge_track1_amd._lib import UrseError
    with pytest.raises(UrseError):
        ops.lstm_fwd_rw(g2, whhb_rw, H, Hp, save=save, paired=True, **sm)
    for tw, paired in ((0, False), (16, False)):
        g2.copy_(gx)
        h2, c2 = ops.lstm_fwd_rw(g2, whhb, H, Hp, save=save, target_wgs=tw, paired=paired, **sm)
        assert torch.equal(h1.view(torch.int16), h2.view(torch.int16))
        assert torch.equal(g1.view(torch.int16), g2.view(torch.int16))
        if save:
            assert torch.equal(c1, c2)
    assert ops.launch_counts()["lstm_fwd_rw"] >= 2


@pytest.mark.parametrize("B,T,K", [(2, 9, 20), (32, 12, 34), (5, 40, 34)])
def test_xcd_aware_clusters_equal_static_clusters(lib, B, T, K):
    """cluster forward with clusters formed from workgroups that read the same XCC id (plain stores for the h hand-off inside an XCD,
    mixed clusters with fewer sequences for the left-over workgroups) == the static clusters, bit for bit: which workgroups form a
    cluster changes where the bytes travel, not one product or sum."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(5)
    N, dev = 196, "cuda"
    H, Hp = 2 * N, 416
    whh = torch.randn(2 * 4 * H, H, device=dev) * 0.05
    whhq = torch.empty(2 * ((H + 3) // 4) * (Hp // 32) * 512, device=dev, dtype=torch.bfloat16)
    ops.call("lstm_pack_quads", whh, whhq, H, Hp, ops.BF16, ops.stream_ptr())
    M = B * T * K
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    assert ops.lstm_cluster_plan(H, Hp, sm["n_seq"]) is not None
    gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
    outs = []
    for xa in (False, True, True):
        g = gx.clone()
        h, c, err = ops.lstm_fwd_cluster(g, whhq, H, Hp, xcd_aware=xa, **sm)
        assert int(err.item()) == 0
        outs.append((g, h, c))
    for o in outs[1:]:
        assert torch.equal(outs[0][0].view(torch.int16), o[0].view(torch.int16))
        assert torch.equal(outs[0][1].view(torch.int16), o[1].view(torch.int16))
        assert torch.equal(outs[0][2], o[2])


def test_cluster_helper_waves_equal_the_fourteen_wave_form_at_full_size(lib, monkeypatch):
    """the time path at C2 (1,088 sequences x 401 steps per direction): the form with two helper waves (saved-gate stores and the next
    step's pre-activations off the working waves) against the 14-wave form, every saved gate, h and c bit for bit.  (A first helper form
    stored 0.03 % of the saved gates with a zeroed first dword - profiles/r04_exp_cluster_helpers_v1.log; only a full-size comparison saw it.)"""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(6)
    N, dev, B, T, K = 196, "cuda", 32, 401, 34
    H, Hp = 2 * N, 416
    whh = torch.randn(2 * 4 * H, H, device=dev) * 0.05
    whhq = torch.empty(2 * ((H + 3) // 4) * (Hp // 32) * 512, device=dev, dtype=torch.bfloat16)
    ops.call("lstm_pack_quads", whh, whhq, H, Hp, ops.BF16, ops.stream_ptr())
    M = B * T * K
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
    outs = []
    for hp in ("0", "2", "2"):
        monkeypatch.setenv("URSE_CLUSTER_HELPERS", hp)
        g = gx.clone()
        h, c, err = ops.lstm_fwd_cluster(g, whhq, H, Hp, **sm)
        assert int(err.item()) == 0
        outs.append((g, h.clone(), c.clone()))
    for o in outs[1:]:
        assert torch.equal(outs[0][0].view(torch.int16), o[0].view(torch.int16))
        assert torch.equal(outs[0][1].view(torch.int16), o[1].view(torch.int16))
        assert torch.equal(outs[0][2], o[2])


@pytest.mark.parametrize("ns,sl,strided", [(300, 34, False), (37, 5, False), (2 * 34, 9, True), (1, 3, False)])
def test_fused_row_wave_forward_matches_two_kernel_form_and_lstm(lib, ns, sl, strided):
    """csrc/lstm_rwx.hip (x W_ih^T + b + h W_hh^T in one accumulator, no gx matrix) against {gate GEMM, lstm_rw} and against
    torch.nn.LSTM in f32: the fused form skips the bf16 rounding of the pre-activation, so it differs from the two-kernel form by
    that rounding and must not be farther from the f32 LSTM than the two-kernel form is."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(6)
    N, dev, dt = 196, "cuda", torch.bfloat16
    H = 2 * N
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
    assert "wx" in pk and ops.lstm_rwx_supported(N, pk["Np"], H, pk["Hp"])
    x = torch.randn(ns, sl, N)
    with torch.no_grad():
        y, _ = lstm(x)
    if strided:       # the time-path row map: sequence s = (b, k), rows (b * sl + t) * K + k
        K, Bb = 34, ns // 34
        rows = x.reshape(Bb, K, sl, N).permute(0, 2, 1, 3).reshape(-1, N)
        yr = y.reshape(Bb, K, sl, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
        sm = dict(n_seq=ns, seq_len=sl, inner=K, outer=sl * K, stride=K)
    else:
        rows, yr = x.reshape(-1, N), y.reshape(-1, 2 * H)
        sm = dict(n_seq=ns, seq_len=sl, inner=1, outer=sl, stride=1)
    M = rows.shape[0]
    xr = ops.pack2d(rows.to(dev), M, pk["Np"], dt)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    h1, c1 = ops.lstm_fwd_rw(gx, pk["whhb"], H, pk["Hp"], **sm)
    g2, h2, c2 = ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], **sm)
    _, h3, _ = ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], save=False, target_wgs=8, **sm)
    assert torch.equal(h2.view(torch.int16), h3.view(torch.int16))
    assert torch.all(h2[:, 2 * H:] == 0)
    e1 = (h1[:, :2 * H].float().cpu() - yr).abs()
    e2 = (h2[:, :2 * H].float().cpu() - yr).abs()
    assert e2.max().item() <= 2e-2 and e2.mean().item() <= 1.1 * e1.mean().item() + 1e-6, (e1.max().item(), e1.mean().item(), e2.max().item(), e2.mean().item())
    assert (h1.float() - h2.float()).abs().max().item() <= 1.6e-2 and (h1.float() - h2.float()).abs().mean().item() <= 5e-4
    assert (gx.float() - g2.float()).abs().max().item() <= 1.6e-2          # saved gate activations
    assert (c1 - c2).abs().max().item() <= 2e-2
    assert ops.launch_counts()["lstm_fwd_rwx"] >= 2


@pytest.mark.parametrize("B,T,K", [(1, 7, 34), (2, 21, 20), (3, 9, 34), (5, 40, 34), (32, 401, 34)])
def test_nsplit_bptt_matches_streaming_kernel(lib, B, T, K):
    """time-path BPTT split over pairs of workgroups by OUTPUT columns (each member streams its half of W_hh^T, the halves of the gate
    gradients are exchanged through the gates output with write-through stores and a flag) vs the one-workgroup streaming kernel: same
    bf16 products; the second member adds its k-slabs in another order."""
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(8)
    N, dev, dt = 196, "cuda", torch.bfloat16
    H = 2 * N
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
    M = B * T * K
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    assert ops.lstm_nsplit_plan(H, sm["n_seq"]) is not None
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dt)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, hout.shape[1], dt)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, rows16=1, **sm)
    _, err = ops.lstm_bwd_nsplit(dh, g2, c, pk["whhT"], H, **sm)
    assert int(err.item()) == 0
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    assert d.max().item() <= 2e-2 * scale and d.mean().item() <= 2e-4 * scale, (d.max().item(), d.mean().item(), scale)
    # the units of member 0 (tiles 0 .. 12) see the k-slabs in the streaming kernel's order: bit-identical gate gradients
    for dr in range(2):
        a = g1[:, dr * 4 * H:dr * 4 * H + 13 * 64].view(torch.int16)
        b = g2[:, dr * 4 * H:dr * 4 * H + 13 * 64].view(torch.int16)
        frac = (a != b).float().mean().item()
        assert frac <= (2e-2 if T < 100 else 5e-2), frac          # (their dh_rec is exact; differences enter through the other half's gradients one step later and pile up over 401 steps: 3.2 %)
    # (round 6: the helper-wave, seven-wave and touch-wave forms - bit-identical to this one when they were tested here in round 5, and slower -
    # are compiled into variant builds only: -DURSE_EXPERIMENTS)


def test_multi_pack_equals_per_lstm_pack(lib):
    """urse_lstm_pack*_multi (one launch per layout for all LSTMs of a model, what every step after the first runs) writes exactly what the
    per-LSTM entry points write, at the C2 widths (N = 196, H = 392), for rows with different optional layouts (time / band path)."""
    from urgent2026_challenge_track1_amd import ops
    N, H, dt = 196, 392, torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    entries, want = [], []
    for i, lay in enumerate(({"whhq", "whhb"}, {"whhq", "whhb", "wx"}, {"wx"})):
        w = [torch.randn(8 * H, N, device="cuda", generator=g), torch.randn(8 * H, H, device="cuda", generator=g),
             torch.randn(8 * H, device="cuda", generator=g), torch.randn(8 * H, device="cuda", generator=g)]
        ref = ops.lstm_pack(*w, N, H, dt, layouts=lay)
        out = {k: (torch.full_like(v, 7) if torch.is_tensor(v) else v) for k, v in ref.items()}
        entries.append(tuple(w) + (out,))
        want.append(ref)
    table = ops.lstm_pack_multi(entries, N, H, dt)
    table2 = ops.lstm_pack_multi(entries, N, H, dt, table=table)
    assert table2[0].data_ptr() == table[0].data_ptr()               # same buffers: the pointer table is reused
    for (_, _, _, _, out), ref in zip(entries, want):
        assert set(out) == set(ref)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(out[k], v), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,K,path", [(2, 9, 20, "time"), (3, 7, 34, "time"), (2, 40, 34, "time"), (2, 40, 34, "band"), (32, 25, 34, "time")])
def test_cluster_forward_with_fused_projection_matches_two_kernel_form(lib, dtype, B, T, K, path):
    """csrc/lstm_clusterx.hip (round 5): x W_ih^T + b + h W_hh^T in ONE kernel on the cluster geometry (7 waves x 2 unit quads) against the gate GEMM +
    lstm_cluster.hip.  Round 5 kept that form's rounding point (the projection rounded to the 16-bit operand format before the recurrent product is added);
    round 6 keeps x W_ih^T + b in f32 until h W_hh^T is added (nn.LSTM's own accumulation: one rounding LESS than the two-kernel form, whose gx matrix is stored
    in 16 bits), so h, c and the saved gate activations agree with it to a 16-bit ulp at most and 1.1e-4 (bf16) / 1.4e-5 (f16) on average; also against nn.LSTM in f32."""
    from urgent2026_challenge_track1_amd import ops
    N, H, dev = 196, 392, "cuda"
    torch.manual_seed(5)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    assert pk.get("wihq") is not None and ops.lstm_clusterx_supported(N, pk["Np"], H, pk["Hp"])
    M = B * T * K
    x = torch.randn(B, T, K, N)
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
        y = lstm(x.permute(0, 2, 1, 3).reshape(B * K, T, N))[0].detach().reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
        y = lstm(x.reshape(B * T, K, N))[0].detach().reshape(-1, 2 * H)
    assert ops.lstm_cluster_plan(H, pk["Hp"], sm["n_seq"]) is not None
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    h1, c1, e1 = ops.lstm_fwd_cluster(gx, pk["whhq"], H, pk["Hp"], **sm)
    g1 = gx.view(torch.bfloat16)
    ops.launch_counts(reset=True)
    g2, h2, c2, e2, h2b = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], bf16_copy=True, **sm)
    assert ops.launch_counts()["lstm_fwd_clusterx"] == 1
    assert int(e1.item()) == 0 and int(e2.item()) == 0
    tol = 2e-2 if dtype == torch.bfloat16 else 2.5e-3
    assert (h2[:, :2 * H].float().cpu() - y).abs().max().item() <= tol
    ulp = 8e-3 if dtype == torch.bfloat16 else 1e-3
    dh, dc, dg = (h1.float() - h2.float()).abs(), (c1 - c2).abs(), (g1.float() - g2.float()).abs()
    print("fused vs two-kernel (%s): h max %.2e mean %.2e, c max %.2e, gates max %.2e" % (dtype, dh.max().item(), dh.mean().item(), dc.max().item(), dg.max().item()))
    # (mean: the fused kernel's pre-activation is not rounded to 16 bits before the recurrent product is added - 1-ulp differences on a few per cent of the bf16 elements)
    assert dh.max().item() <= 2 * ulp and dh.mean().item() <= (3e-4 if dtype == torch.bfloat16 else 1e-4) and dc.max().item() <= 4 * ulp and dg.max().item() <= 2e-2
    assert torch.all(h2[:, 2 * H:] == 0) and g2.dtype == torch.bfloat16
    if dtype == torch.float16:
        assert torch.equal(h2b[:, :2 * H], h2[:, :2 * H].float().to(torch.bfloat16)) and torch.all(h2b[:, 2 * H:] == 0)
    # inference form: nothing saved
    g3, h3, c3, e3 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=False, **sm)
    assert g3 is None and c3 is None and torch.equal(h3, h2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_seq,K", [(2400, 34), (1300, 5), (1153, 8), (1300, 1), (2305, 2), (12832, 34)])
def test_cluster_forward_in_rounds_on_the_band_path(lib, dtype, n_seq, K):
    """Round 6: the band path (espnet2 BSRNN's rnn_band, reference twin baseline_code/models/bsrnn_flowse.py:296-299: many short sequences) through the
    fused cluster forward in ROUNDS - every co-resident cluster keeps its weights and takes 64 sequences per round; the hand-off's step counter runs on
    across rounds, a round's first step waits for the previous round's last publication and starts from h = 0.  Checked (a) against nn.LSTM in f32,
    (b) BIT for bit against the same kernel run one round at a time on slabs of at most clusters x 64 sequences (a sequence's arithmetic does not
    depend on which cluster, round or row it gets), incl. a last round that only some clusters take part in, a partial last chunk, an odd sequence
    length (the LDS tiles' parity restarts per round, the exchange planes' does not) and the full C2 shape (12 rounds); (c) forward-only form."""
    from urgent2026_challenge_track1_amd import ops
    N, H, dev = 196, 392, "cuda"
    torch.manual_seed(11)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    plan = ops.lstm_clusterx_plan(H, pk["Hp"], n_seq)
    cap = plan[1] * 64
    assert plan[6] == -(-n_seq // cap) and plan[6] > 1 and ops.lstm_cluster_plan(H, pk["Hp"], n_seq) is None
    M = n_seq * K
    x = torch.randn(n_seq, K, N)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
    sm = dict(n_seq=n_seq, seq_len=K, inner=1, outer=K, stride=1)
    ops.launch_counts(reset=True)
    g, h, c, e = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], **sm)
    assert ops.launch_counts()["lstm_fwd_clusterx"] == 1 and int(e.item()) == 0
    if n_seq <= 2400:
        y = lstm(x)[0].detach().reshape(-1, 2 * H)
        assert (h[:, :2 * H].float().cpu() - y).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 2.5e-3)
    for s0 in range(0, n_seq, cap):
        s1 = min(n_seq, s0 + cap)
        sl = slice(s0 * K, s1 * K)
        g1, h1, c1, e1 = ops.lstm_fwd_clusterx(xr[sl], pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], n_seq=s1 - s0, seq_len=K, inner=1, outer=K, stride=1)
        assert int(e1.item()) == 0
        assert torch.equal(h1, h[sl]) and torch.equal(c1, c[sl]) and torch.equal(g1, g[sl]), (s0, s1)
    g3, h3, c3, e3 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=False, **sm)
    assert g3 is None and c3 is None and torch.equal(h3, h) and int(e3.item()) == 0
    if dtype == torch.float16 and n_seq <= 2400:      # the instance that writes h once more in bf16 (f16-forward training without the mixed-operand weight gradients)
        g4, h4, c4, e4, h4b = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], bf16_copy=True, **sm)
        assert int(e4.item()) == 0 and torch.equal(h4, h) and torch.equal(g4, g) and torch.equal(c4, c)
        assert torch.equal(h4b[:, :2 * H], h[:, :2 * H].float().to(torch.bfloat16)) and torch.all(h4b[:, 2 * H:] == 0)


def test_cluster_forward_in_rounds_on_the_time_geometry(lib):
    """the same rounds with the time path's strides (sequence (b, k), step stride K): 1,360 sequences of 7 steps = 2 rounds, against nn.LSTM."""
    from urgent2026_challenge_track1_amd import ops
    N, H, dev, dtype = 196, 392, "cuda", torch.bfloat16
    B, T, K = 40, 7, 34
    torch.manual_seed(12)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    M = B * T * K
    x = torch.randn(B, T, K, N)
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    assert ops.lstm_clusterx_plan(H, pk["Hp"], sm["n_seq"])[6] == 2
    y = lstm(x.permute(0, 2, 1, 3).reshape(B * K, T, N))[0].detach().reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
    g, h, c, e = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], **sm)
    assert int(e.item()) == 0
    assert (h[:, :2 * H].float().cpu() - y).abs().max().item() <= 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,K,path,nt", [(2, 30, 34, "time", 1), (12, 12, 34, "time", 2), (1, 250, 34, "band", 1), (7, 401, 34, "time", 1), (8, 40, 34, "time", 2), (14, 9, 34, "time", 2), (16, 9, 34, "time", 3), (21, 5, 34, "time", 3)])
def test_cluster_forward_instances_of_fewer_row_tiles(lib, monkeypatch, dtype, B, T, K, path, nt):
    """Round 6: launches whose plan gives a cluster at most 16 / 32 sequences (small batches: the time path of B <= 8 / 16 utterances at 48 kHz, where an inference
    forward is 401 steps of hand-off latency) run instances of the fused cluster forward that gather, fetch, multiply and store ONE / TWO / THREE row tiles of 16 per step
    (template parameter NT).  A sequence's arithmetic is the same in every instance: h, c and the saved gates are bit-identical to the full instance (forced with
    URSE_CLUSTERX_NT=4), with and without saving."""
    from urgent2026_challenge_track1_amd import ops
    N, H, dev = 196, 392, "cuda"
    torch.manual_seed(13)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    M = B * T * K
    x = torch.randn(B, T, K, N)
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K) if path == "time" else dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    plan = ops.lstm_clusterx_plan(H, pk["Hp"], sm["n_seq"])
    bound = -(-sm["n_seq"] // (plan[1] - 2))      # the instance is chosen for the rows a cluster can get once the XCD-aware formation has left two clusters per direction empty
    assert plan[6] == 1 and 16 * (nt - 1) < bound <= 16 * nt, (plan, bound)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
    monkeypatch.delenv("URSE_CLUSTERX_NT", raising=False)
    g1, h1, c1, e1 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], **sm)
    _, h1i, _, e1i = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=False, **sm)
    monkeypatch.setenv("URSE_CLUSTERX_NT", "4")
    g4, h4, c4, e4 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], **sm)
    assert int(e1.item()) == 0 and int(e1i.item()) == 0 and int(e4.item()) == 0
    assert torch.equal(h1, h4) and torch.equal(c1, c4) and torch.equal(g1, g4) and torch.equal(h1i, h4)
    if M <= 40000:
        if path == "time":
            y = lstm(x.permute(0, 2, 1, 3).reshape(B * K, T, N))[0].detach().reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
        else:
            y = lstm(x.reshape(B * T, K, N))[0].detach().reshape(-1, 2 * H)
        assert (h1[:, :2 * H].float().cpu() - y).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 2.5e-3)
