"""BSRNN-Flow (BASELINE.json configs[3]) at its REAL width - N = 384, H = 768, L = 6, F = 769 bins, K = 48 bands, 103,245,488 parameters,
the shapes `bench.py --model flow` times - against oracle/flow_ref.py, which tests/golden/make_golden_flow.py pins bit-exact to the reference's
own DNN (baseline_code/models/bsrnn_flowse.py:255-318; loss: flow_model.py:149-172).  tests/test_flow_gpu.py compares at N = 16, L = 2 only.
  f32 mode (exact-f32 MFMA): DNN output, flow-matching loss and EVERY parameter gradient within 1e-3;
  bf16 mode (the benchmarked dispatch: cluster2 forward, split BPTT): launch counters prove those kernels ran; errors bounded and logged.
B = 2 x 1 s @ 48 kHz (T = 126 frames; the benchmark's batch, a quarter of its length: the dispatch depends on the number of sequences -
96 on the time path - and on H, not on the length)."""
import pytest
import torch

from oracle import flow_ref
from tests import parity_log

pytestmark = pytest.mark.gpu

N, L, B, FS, SECONDS = 384, 6, 2, 48000, 1.0


def _setup(seed=3):
    torch.manual_seed(seed)
    ref = flow_ref.FlowSE(bsrnn_hidden=N, num_layer=L)
    with torch.no_grad():
        for n, p in ref.dnn.named_parameters():
            if "norm" in n and p.requires_grad:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(seed + 1)
    Ls = int(SECONDS * FS)
    clean = 0.2 * torch.randn(B, Ls, generator=g)
    noisy = clean + 0.05 * torch.randn(B, Ls, generator=g)
    lens = torch.full((B,), Ls)
    x0 = ref.speech_to_feature(clean, FS, lens)
    y = ref.speech_to_feature(noisy, FS, lens)
    t = torch.tensor([0.7, 0.2])
    z = torch.randn(x0.shape, dtype=torch.complex64, generator=g)
    return ref, x0, y, t, z


def _oracle(ref, x0, y, t, z):
    mean, std = ref.ode.marginal_prob(x0, t, y)
    xt = mean + std[:, None, None, None] * z
    out = ref.dnn(torch.cat([xt, y], dim=1), t)                       # [B, 1, F, T] complex
    loss = ref.loss_from(x0, y, t, z)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in ref.dnn.named_parameters() if p.requires_grad}
    return out.detach(), float(loss), grads


def _ri(c):      # complex [B, 1, F, T] -> f32 [B, T, F, 2] on the GPU
    return torch.view_as_real(c.squeeze(1).permute(0, 2, 1).contiguous()).cuda()


def _gpu(dtype, ref, x0, y, t, z):
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd._lib import call, stream_ptr
    from urgent2026_challenge_track1_amd.config import Config
    from urgent2026_challenge_track1_amd.flow_model import FlowSEModel, _FlowLossFn
    m = FlowSEModel(Config(bsrnn_hidden=N, num_layer=L, compute_dtype=dtype, sigma_min=0.05, sigma_max=0.5))
    m.dnn.load_state_dict(ref.dnn.state_dict(), strict=True)
    m = m.cuda()
    x_ri, y_ri, z_ri = _ri(x0), _ri(y), _ri(z)
    xt, cvf = torch.empty_like(x_ri), torch.empty_like(x_ri)
    call("flow_prepare", x_ri, y_ri, z_ri, t.cuda(), xt, cvf, B, xt[0].numel() // 2, 0.05, 0.5, stream_ptr())
    ops.launch_counts(reset=True)
    with torch.no_grad():
        out = m.dnn(xt, y_ri, t.cuda(), sign=1.0)
    vf = m.vector_field_ri(xt, t.cuda(), y_ri)
    loss = _FlowLossFn.apply(vf, cvf)
    loss.backward()
    m.dnn._flush_deferred_wgrads()
    torch.cuda.synchronize()
    ops.poll_kernel_errors(torch.device("cuda", torch.cuda.current_device()), sync=True)
    return m, out, float(loss), ops.launch_counts()


def _compare(m, out, loss, ref_out, ref_loss, ref_grads):
    ro = _ri(ref_out).cpu()
    e_out = float((out.cpu() - ro).abs().max() / ro.abs().max())
    l2_out = float((out.cpu() - ro).norm() / ro.norm())
    e_loss = abs(loss - ref_loss) / abs(ref_loss)
    mine = dict(m.dnn.named_parameters())
    worst_max, worst_l2, wn = 0.0, 0.0, None
    for n, gr in ref_grads.items():
        g = mine[n].grad.cpu()
        e = float((g - gr).abs().max() / (gr.abs().max() + 1e-30))
        r = float((g - gr).norm() / (gr.norm() + 1e-30))
        if r > worst_l2:
            worst_l2, wn = r, n
        worst_max = max(worst_max, e)
    return dict(out_max_over_peak=e_out, out_rel_l2=l2_out, loss_rel=e_loss, worst_grad_max_over_peak=worst_max,
                worst_grad_rel_l2=worst_l2, worst_grad=wn)


@pytest.fixture(scope="module")
def c4_oracle():
    ref, x0, y, t, z = _setup()
    return (ref, x0, y, t, z) + _oracle(ref, x0, y, t, z)


def test_c4_full_width_f32_matches_oracle_1e3(lib, c4_oracle):
    ref, x0, y, t, z, ref_out, ref_loss, ref_grads = c4_oracle
    m, out, loss, counts = _gpu("f32", ref, x0, y, t, z)
    fig = _compare(m, out, loss, ref_out, ref_loss, ref_grads)
    print("C4 full width, f32:", fig)
    parity_log.record("c4_fullwidth_f32_vs_oracle", shape="B2 x 1 s @ 48 kHz, N=384, L=6, F=769", **fig)
    assert fig["out_max_over_peak"] <= 1e-3 and fig["out_rel_l2"] <= 1e-3 and fig["loss_rel"] <= 1e-3, fig
    assert fig["worst_grad_rel_l2"] <= 1e-3 and fig["worst_grad_max_over_peak"] <= 2e-3, fig


def test_c4_full_width_bf16_dispatch_bounded(lib, c4_oracle):
    ref, x0, y, t, z, ref_out, ref_loss, ref_grads = c4_oracle
    m, out, loss, counts = _gpu("bf16", ref, x0, y, t, z)
    fig = _compare(m, out, loss, ref_out, ref_loss, ref_grads)
    print("C4 full width, bf16:", fig, {k: v for k, v in counts.items() if v})
    parity_log.record("c4_fullwidth_bf16_vs_f32_oracle", shape="B2 x 1 s @ 48 kHz, N=384, L=6, F=769",
                      launch_counts={k: v for k, v in counts.items() if v}, **fig)
    # the kernels `bench.py --model flow` times: H = 768 cluster forward (time path: one launch, band path: row-block launches) and the split BPTT
    assert counts["lstm_fwd_cluster2"] > 0 and counts["lstm_bwd_split"] > 0, counts
    assert counts["lstm_fwd_stream"] == 0, counts
    # 8-bit operand mantissas through 6 layers; bounds = 2x the figures observed on the GPU (profiles/r05_c2_parity_v1.json: output 4.7e-3 rel. L2 /
    # 4.6e-3 of the peak, loss 5e-6, worst gradient 1.14e-2 rel. L2)
    assert fig["out_rel_l2"] <= 9.4e-3 and fig["out_max_over_peak"] <= 9.2e-3 and fig["loss_rel"] <= 1e-4 and fig["worst_grad_rel_l2"] <= 2.3e-2, fig


# ---------------------------------------------------------------------------------------------------------------------------------------
# the sampler's OUTPUT WAVEFORM at full width, and one backward at the length the bench trains on
# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c4_sampler_oracle():
    """FlowSEModel.enhance (flow_model.py:189-200; sampling/__init__.py:46-60, odesolvers.py:72-81) on the oracle: N = 384, L = 6, one 1 s
    utterance @ 48 kHz, 15 Euler steps from a fixed complex z."""
    torch.manual_seed(11)
    ref = flow_ref.FlowSE(bsrnn_hidden=N, num_layer=L)
    with torch.no_grad():
        for n, p in ref.dnn.named_parameters():
            if "norm" in n and p.requires_grad:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(12)
    Ls = int(SECONDS * FS)
    noisy = 0.2 * torch.randn(1, Ls, generator=g) + 0.05 * torch.randn(1, Ls, generator=g)
    lens = torch.tensor([Ls])
    Y = ref.speech_to_feature(noisy, FS, lens)
    z = torch.randn(Y.shape, dtype=torch.complex64, generator=g)
    x = ref.enhance_from(Y, z, N=15)
    wav = ref.feature_to_speech(x, FS, lens)
    return ref, noisy, lens, z, x, wav


def _enhance_gpu(dtype, ref, noisy, lens, z):
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd.config import Config
    from urgent2026_challenge_track1_amd.flow_model import FlowSEModel
    m = FlowSEModel(Config(bsrnn_hidden=N, num_layer=L, compute_dtype=dtype, sigma_min=0.05, sigma_max=0.5))
    m.dnn.load_state_dict(ref.dnn.state_dict(), strict=True)
    m = m.cuda()
    ops.launch_counts(reset=True)
    with torch.no_grad():
        Y = m.speech_to_feature_ri(noisy.cuda(), FS, lens)
        x = m.sample_ri(Y, 15, z_ri=_ri(z))
        wav = m.feature_ri_to_speech(x, FS, lens)
    torch.cuda.synchronize()
    return x.cpu(), wav.cpu(), {k: v for k, v in ops.launch_counts().items() if v}


def _wave_figures(x, wav, ref_x, ref_wav):
    rx = _ri(ref_x).cpu()
    return dict(wave_rel_l2=float((wav - ref_wav).norm() / ref_wav.norm()),
                wave_max_over_peak=float((wav - ref_wav).abs().max() / ref_wav.abs().max()),
                feature_rel_l2=float((x - rx).norm() / rx.norm()))


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_c4_full_width_enhance_waveform_vs_oracle(lib, c4_sampler_oracle, dtype):
    """enhance = STFT + 15 chained DNN evaluations + iSTFT.  f32 mode and f16 operands: the enhanced waveform within north_star's 1e-3
    of the oracle's; bf16 operands recorded (8-bit mantissas through 15 x 6 layers), bounded at twice what was observed."""
    ref, noisy, lens, z, ref_x, ref_wav = c4_sampler_oracle
    x, wav, counts = _enhance_gpu(dtype, ref, noisy, lens, z)
    fig = _wave_figures(x, wav, ref_x, ref_wav)
    print("C4 enhance, full width,", dtype, fig, counts)
    parity_log.record("c4_fullwidth_enhance_%s_vs_oracle" % dtype, shape="1 x 1 s @ 48 kHz, N=384, L=6, F=769, Euler N=15",
                      launch_counts=counts, **fig)
    assert torch.isfinite(wav).all()
    if dtype != "f32":
        assert counts.get("lstm_fwd_cluster2", 0) > 0, counts           # the kernel `flow_c4.enhance_ms` times
    if dtype in ("f32", "f16"):
        assert fig["wave_rel_l2"] <= 1e-3 and fig["wave_max_over_peak"] <= 1e-3, fig
    else:
        assert fig["wave_rel_l2"] <= 2e-2 and fig["wave_max_over_peak"] <= 2e-2, fig


def test_zz_c4_backward_at_the_benchmarked_length_bf16(lib):
    """`flow_c4` trains on 4 s utterances: T = 501 frames, a 501-step BPTT at H = 768 on the time path (cooperative split kernel).  One
    forward + backward at B 1 x 4 s in the benchmarked bf16 dispatch against the f32 oracle's backward (the oracle needs minutes of host
    time: the test is named to run last in this file)."""
    torch.manual_seed(21)
    ref = flow_ref.FlowSE(bsrnn_hidden=N, num_layer=L)
    with torch.no_grad():
        for n, p in ref.dnn.named_parameters():
            if "norm" in n and p.requires_grad:
                p.add_(0.1 * torch.randn_like(p))
    g = torch.Generator().manual_seed(22)
    Ls = 4 * FS
    clean = 0.2 * torch.randn(1, Ls, generator=g)
    noisy = clean + 0.05 * torch.randn(1, Ls, generator=g)
    lens = torch.tensor([Ls])
    x0, y = ref.speech_to_feature(clean, FS, lens), ref.speech_to_feature(noisy, FS, lens)
    assert x0.shape[-1] == 501
    t = torch.tensor([0.6])
    z = torch.randn(x0.shape, dtype=torch.complex64, generator=g)
    ref_out, ref_loss, ref_grads = _oracle(ref, x0, y, t, z)
    global B
    keep, B = B, 1
    try:
        m, out, loss, counts = _gpu("bf16", ref, x0, y, t, z)
    finally:
        B = keep
    fig = _compare(m, out, loss, ref_out, ref_loss, ref_grads)
    print("C4 full width, B1 x 4 s (T = 501), bf16:", fig, {k: v for k, v in counts.items() if v})
    parity_log.record("c4_fullwidth_T501_bf16_vs_f32_oracle", shape="B1 x 4 s @ 48 kHz (T = 501), N=384, L=6, F=769",
                      launch_counts={k: v for k, v in counts.items() if v}, **fig)
    assert counts["lstm_fwd_cluster2"] > 0 and counts["lstm_bwd_split"] > 0, counts
    # bounds: twice the 1 s figures of test_c4_full_width_bf16_dispatch_bounded (four times the recurrent steps on the time path)
    assert fig["out_rel_l2"] <= 1.9e-2 and fig["loss_rel"] <= 2e-4 and fig["worst_grad_rel_l2"] <= 4.6e-2, fig
