"""GPU parity of the f16 FORWARD mode (compute_dtype "f16": IEEE-half operands - 11 significant bits at bf16's bytes and MFMA rate - in every
forward contraction, bf16 operands in the backward): each kernel that takes URSE_F16 against float64 arithmetic on the same f16-rounded
operands, the bf16 copies the backward reads, and the whole model against the f32 oracle.  Why the mode exists: bf16 operands put the enhanced
waveform 4e-3 from the f32 reference arithmetic (north_star asks for 1e-3); the CPU experiment tests/exp_f16_emulation.py
(profiles/r05_exp_f16_emulation_v1.log) showed f16 operands reach 5e-4 and that the band split / mask decoder roundings, not the recurrences,
dominate.  Reference arithmetic: baseline_code/models/bsrnn.py:36-41 (espnet2 BSRNN, torch f32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
f16, bf16 = torch.float16, torch.bfloat16


def _mk(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


@pytest.mark.parametrize("M,N,K,kind", [(300, 200, 224, "nt_128"), (77, 196, 800, "nt_128"), (2500, 196, 800, "nt_ring"), (2309, 790, 224, "nt_ring"),
                                        (16500, 3136, 224, "nt_bres"), (9000, 784, 224, "nt_bres")])
def test_gemm_nt_f16(lib, M, N, K, kind):
    from urgent2026_challenge_track1_amd import ops
    A, W = _mk((M, K), f16, 1), _mk((N, K), f16, 2)
    bias = _mk((N,), torch.float32, 3)
    ref = A.double() @ W.double().T + bias.double()
    ops.launch_counts(reset=True)
    if kind != "nt_bres":
        got = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), out_dtype=torch.float32).cpu()
        assert (got.double() - ref).abs().max().item() <= 2e-5 * (K ** 0.5) * 4 * max(1.0, ref.abs().max().item() / 10)
        res = _mk((M, N), torch.float32, 4)
        r = res.clone().cuda()
        ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), resid=r, out=r)
        assert (r.cpu().double() - (ref + res.double())).abs().max().item() <= 2e-5 * (K ** 0.5) * 4 * max(1.0, ref.abs().max().item() / 10)
    # 16-bit output: f16 (2^-11 relative), with tanh and the bf16 copy of the same values (the mask decoder's hidden layer)
    got16 = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), out_dtype=f16).cpu()
    assert got16.dtype == f16
    assert ((got16.double() - ref).abs() <= 6e-4 * ref.abs() + 2e-3).all()
    c2 = torch.full((M, N), float("nan"), dtype=bf16, device="cuda")
    th = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), act=1, out_dtype=f16, resid=c2).cpu()
    assert (th.double() - torch.tanh(ref)).abs().max().item() <= 1.5e-3
    assert (c2.cpu().double() - th.double()).abs().max().item() <= 4e-3 and torch.isfinite(c2.float()).all()    # (with the second output the call takes the ring / 128 kernel)
    assert ops.launch_counts()[kind] > 0, ops.launch_counts()


def test_gemm_nt_grouped_f16_with_bf16_copy(lib, monkeypatch):
    """per-band records on the ring kernel and on the 128 x 128 kernel: f16 operands, tanh, f16 output + its bf16 copy through the aux slot"""
    import numpy as np
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd.bsrnn import nt_grouped, _ptr
    for min_m, kind in (("512", "nt_grouped_ring"), ("100000", "nt_grouped_128")):
        monkeypatch.setenv("URSE_NT_GROUPED_MIN_M", min_m)
        ops.launch_counts(reset=True)
        M, K, Ns = 700, 224, (784, 200, 784)
        rows, keep, refs = [], [], []
        for g, N in enumerate(Ns):
            A, W = _mk((M, K), f16, 10 + g).cuda(), _mk((N, K), f16, 20 + g).cuda()
            b = _mk((N,), torch.float32, 30 + g).cuda()
            C = torch.empty(M, N + 16, dtype=f16, device="cuda")
            C2 = torch.full((M, N + 16), float("nan"), dtype=bf16, device="cuda")
            rows.append([_ptr(A), _ptr(W), _ptr(C), _ptr(b), _ptr(C2), K, K, N + 16, M, N, K, N + 16])
            keep.append((A, W, b, C, C2))
            refs.append(torch.tanh(A.double().cpu() @ W.double().cpu().T + b.double().cpu()))
        nt_grouped(rows, "cuda", ops.F16, ops.F16, act=1)
        assert ops.launch_counts()[kind] == 1, ops.launch_counts()
        for (A, W, b, C, C2), ref, N in zip(keep, refs, Ns):
            assert (C[:, :N].cpu().double() - ref).abs().max().item() <= 1.5e-3
            assert (C2[:, :N].cpu().double() - C[:, :N].cpu().double()).abs().max().item() <= 4e-3
            assert torch.isnan(C2[:, N:].float()).all()          # nothing written outside the N columns


def test_groupnorm_and_bandsplit_f16_outputs_and_bf16_copies(lib):
    from urgent2026_challenge_track1_amd import ops
    B, T, K, N, Np = 2, 50, 6, 196, 224
    x = torch.randn(B, T, K, N, device="cuda")
    gam, bet = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    y32, st = ops.groupnorm_fwd(x, gam, bet, B, T, 1, K * N, N, Np, 0, torch.float32)
    y16, st2, yb = ops.groupnorm_fwd(x, gam, bet, B, T, 1, K * N, N, Np, 0, f16, bf16_copy=True)
    assert y16.dtype == f16 and yb.dtype == bf16 and torch.equal(st, st2)
    assert torch.equal(y16, y32.to(f16)) and torch.equal(yb, y32.to(bf16))          # one rounding each, from the same f32 value
    assert torch.all(y16[:, N:] == 0) and torch.all(yb[:, N:] == 0)
    ya, _ = ops.groupnorm_fwd(x, gam, bet, B, T, 1, K * N, N, Np, 0, f16, stats=st)
    assert torch.equal(ya, y16)


def _lstm(N, seed, dtype):
    from urgent2026_challenge_track1_amd import ops
    torch.manual_seed(seed)
    H, dev = 2 * N, "cuda"
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    return lstm, pk


@pytest.mark.parametrize("path", ["time", "band"])
@pytest.mark.parametrize("B,T,K,N", [(2, 9, 20, 16), (3, 7, 34, 196), (2, 40, 34, 196)])
def test_lstm_forward_f16_streaming_and_cluster(lib, path, B, T, K, N):
    """f16 streaming forward vs nn.LSTM (f32): 8x closer than the bf16 kernel's bound; f16 cluster forward == f16 streaming; saved gates are bf16;
    the bf16 copy of h is the f16 h rounded once more; the backward runs on them unchanged."""
    from urgent2026_challenge_track1_amd import ops
    H, dev = 2 * N, "cuda"
    lstm, pk = _lstm(N, 1, f16)
    assert pk["wih"].dtype == f16 and pk["whh"].dtype == f16 and pk["wihT"].dtype == bf16 and pk["whhT"].dtype == bf16 and pk["whhq"].dtype == f16
    M = B * T * K
    x = torch.randn(B, T, K, N)
    if path == "time":
        sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
        seqs = x.permute(0, 2, 1, 3).reshape(B * K, T, N)
        y = lstm(seqs)[0].detach().reshape(B, K, T, 2 * H).permute(0, 2, 1, 3).reshape(-1, 2 * H)
    else:
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
        y = lstm(x.reshape(B * T, K, N))[0].detach().reshape(-1, 2 * H)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], f16)
    gx1 = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    assert gx1.dtype == f16
    gx2 = gx1.clone()
    h1, c1, h1b = ops.lstm_fwd(gx1, pk["whh"], H, pk["Hp"], bf16_copy=True, **sm)
    err = (h1[:, :2 * H].float().cpu() - y).abs().max().item()
    assert err <= 2.5e-3, err                                   # (bf16 kernel: 2e-2)
    assert h1.dtype == f16 and h1b.dtype == bf16 and torch.equal(h1b[:, :2 * H], h1[:, :2 * H].float().to(bf16))
    g1 = gx1.view(bf16)
    assert torch.isfinite(g1.float()).all() and g1.float().abs().max().item() <= 1.0          # gate ACTIVATIONS, bf16
    assert ops.lstm_cluster_plan(H, pk["Hp"], sm["n_seq"]) is not None
    h2, c2, err_flag, h2b = ops.lstm_fwd_cluster(gx2, pk["whhq"], H, pk["Hp"], bf16_copy=True, **sm)
    assert int(err_flag.item()) == 0
    assert (h1.float() - h2.float()).abs().max().item() <= 2e-3 and (c1 - c2).abs().max().item() <= 4e-3
    assert (g1.float() - gx2.view(bf16).float()).abs().max().item() <= 2e-2
    assert torch.equal(h2b[:, :2 * H], h2[:, :2 * H].float().to(bf16)) and torch.all(h2b[:, 2 * H:] == 0)
    # the BPTT takes the saved state as it takes the bf16 forward's
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev), M, h1.shape[1], bf16)
    dg = ops.lstm_bwd(dh, g1.clone(), c1, pk["whhT"], H, **sm)
    assert torch.isfinite(dg.float()).all()


def test_lstm_fused_rowwave_f16(lib):
    """fused row-wave forward (input projection inside the recurrence) in f16 against nn.LSTM f32 and against the bf16 form"""
    from urgent2026_challenge_track1_amd import ops
    N, H, B, T, K, dev = 196, 392, 2, 60, 34, "cuda"
    errs = {}
    for dt in (f16, bf16):
        lstm, pk = _lstm(N, 3, dt)
        torch.manual_seed(7)
        x = torch.randn(B * T, K, N)
        y = lstm(x)[0].detach().reshape(-1, 2 * H)
        xr = ops.pack2d(x.reshape(-1, N).to(dev), B * T * K, pk["Np"], dt)
        sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
        gates, h, c, hb = ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], bf16_copy=True, **sm)
        assert gates.dtype == bf16 and h.dtype == dt and hb.dtype == bf16
        errs[dt] = ((h[:, :2 * H].float().cpu() - y).abs().max().item(), (h[:, :2 * H].float().cpu() - y).abs().mean().item())
        if dt == f16:
            assert torch.equal(hb[:, :2 * H], h[:, :2 * H].float().to(bf16)) and torch.all(hb[:, 2 * H:] == 0)
        gi, _, _ = ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], save=False, **sm)
        assert gi is None
    print("fused row-wave forward vs nn.LSTM: f16 max %.2e mean %.2e, bf16 max %.2e mean %.2e" % (errs[f16] + errs[bf16]))
    assert errs[f16][0] <= 2.5e-3 and errs[f16][1] <= 0.25 * errs[bf16][1]


def _c2_models(L, dtype, seed=0, N=196):
    from oracle import bsrnn_ref
    from urgent2026_challenge_track1_amd.config import Config
    from urgent2026_challenge_track1_amd.d_model import SEModel
    torch.manual_seed(seed)
    ref = bsrnn_ref.BSRNN_SE(N, L)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    model = SEModel(Config(model_configs={"num_channel": N, "num_layer": L}, compute_dtype=dtype))
    model.se_model.load_state_dict(ref.state_dict())
    return ref, model.cuda()


@pytest.mark.parametrize("fs,B,Ls", [(48000, 6, 48000), (16000, 6, 16000), (48000, 8, 48480)])
def test_f16_c2_kernel_set_forward_and_gradients(lib, monkeypatch, fs, B, Ls):
    """N = 196, L = 6, B 6 x 1 s, the C2 dispatch (thresholds on the NUMBER of band-path sequences lowered as in tests/test_c2_parity_gpu.py):
    f16 forward against the f32 oracle within north_star's 1e-3 on waveform and loss; gradients (bf16 backward on the f16 forward's state)
    no worse than the bf16 mode's bound."""
    from oracle import losses_ref
    from tests import parity_log
    from urgent2026_challenge_track1_amd import ops
    monkeypatch.setattr(ops, "RW_MIN_SEQ", 1)
    monkeypatch.setattr(ops, "BAND_PATH_NO_CLUSTER", True)
    monkeypatch.setenv("URSE_NT_GROUPED_MIN_M", "512")
    monkeypatch.setattr(ops, "BWD_ROWS16", {"f": 2 | 16})
    ref, model = _c2_models(6, "f16")
    g = torch.Generator().manual_seed(1)
    # B 8 x 1.01 s (T = 102 frames, M = B T K = 27,744 rows = whole 32-row stages): the weight gradients take the MIXED-operand kernels (round 6:
    # bf16 gradients against the forward's f16 x_n / h, no second bf16 copy); B 6 x 1 s (M = 20,604): the two-copy form
    mixed = (B, Ls) == (8, 48480)
    clean = 0.3 * torch.randn(B, Ls, generator=g)
    noisy = clean + 0.1 * torch.randn(B, Ls, generator=g)
    lens = torch.full((B,), Ls, dtype=torch.int32)
    lens[1] = Ls - Ls // 16
    wav_r, _ = ref(noisy, lens, fs, False)
    loss_r = losses_ref.mr_l1_loss(clean, wav_r).mean()
    loss_r.backward()
    ops.launch_counts(reset=True)
    wav = model.se_model(noisy.cuda(), lens, fs)[0]
    loss = ops.mr_l1_loss(clean.cuda(), wav).mean()
    loss.backward()
    model.se_model.core._flush_deferred_wgrads()
    torch.cuda.synchronize()
    counts = ops.launch_counts()
    for k in ("nt_bres", "nt_ring", "lstm_fwd_clusterx", "nt_grouped_ring", "lstm_bwd_nsplit"):
        assert counts[k] > 0, (k, counts)
    assert counts["tn_dual"] > 0 or fs != 48000, counts       # (K = 27 bands at 16 kHz: the dual weight-gradient kernel's whole-block condition does not hold)
    # 12 half layers x (2 dual launches + 1 fc gradient) in the mixed form, or none of them
    assert counts["tn_act_f16"] == (36 if mixed else 0), counts
    assert counts["lstm_fwd_stream"] == 0 and counts["nt_128"] == 0, counts
    wc, wr = wav.detach().cpu(), wav_r.detach()
    l2 = float((wc - wr).norm() / wr.norm())
    mx = float((wc - wr).abs().max() / wr.abs().max())
    el = abs(float(loss) - float(loss_r)) / abs(float(loss_r))
    mine = dict(model.se_model.named_parameters())
    worst, wn = 0.0, None
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert torch.all(mine[n].grad == 0), n
            continue
        r = float((mine[n].grad.cpu() - p.grad).norm() / (p.grad.norm() + 1e-30))
        if r > worst:
            worst, wn = r, n
    print("f16 C2 kernel set @ %d Hz: wav rel. L2 %.2e, max / peak %.2e, loss %.2e, worst grad rel. L2 %.2e (%s)" % (fs, l2, mx, el, worst, wn))
    parity_log.record("f16_c2_kernel_set_L6_fs%d%s" % (fs, "_mixed_wgrad" if mixed else ""), shape="B%d x %.2f s, N=196" % (B, Ls / fs), wav_rel_l2=l2, wav_max_over_peak=mx, loss_rel=el, worst_grad_rel_l2=worst,
                      worst_grad=wn)
    assert l2 <= 1e-3 and mx <= 1e-3 and el <= 1e-3, (l2, mx, el)
    assert worst <= 2.1e-2, (worst, wn)


def test_f16_small_model_all_rates_and_inference_path(lib):
    """small shapes take the generic kernels (128 x 128 GEMMs, streaming recurrence): forward + every gradient finite and close to the f32 oracle at
    several corpus rates; eval (no saved state, no bf16 copies) equals the training forward bit for bit"""
    from oracle import losses_ref
    from urgent2026_challenge_track1_amd import ops
    ref, model = _c2_models(2, "f16", seed=3, N=16)
    for fs in (8000, 22050, 48000):
        g = torch.Generator().manual_seed(fs)
        Ls = fs // 2
        clean = 0.3 * torch.randn(2, Ls, generator=g)
        noisy = clean + 0.1 * torch.randn(2, Ls, generator=g)
        lens = torch.tensor([Ls, Ls - 100], dtype=torch.int32)
        ref.zero_grad(set_to_none=True)
        wav_r, _ = ref(noisy, lens, fs, False)
        loss_r = losses_ref.mr_l1_loss(clean, wav_r).mean()
        loss_r.backward()
        model.se_model.core.flat_grads.zero_()
        wav = model.se_model(noisy.cuda(), lens, fs)[0]
        loss = ops.mr_l1_loss(clean.cuda(), wav).mean()
        loss.backward()
        model.se_model.core._flush_deferred_wgrads()
        with torch.no_grad():
            wav_e = model.se_model(noisy.cuda(), lens, fs)[0]
        assert torch.equal(wav_e, wav.detach())
        wr = wav_r.detach()
        assert float((wav.detach().cpu() - wr).norm() / wr.norm()) <= 1e-3
        assert abs(float(loss) - float(loss_r)) <= 1e-3 * abs(float(loss_r))
        mine = dict(model.se_model.named_parameters())
        for n, p in ref.named_parameters():
            if p.grad is not None:
                r = float((mine[n].grad.cpu() - p.grad).norm() / (p.grad.norm() + 1e-30))
                assert r <= 3e-2, (fs, n, r)
