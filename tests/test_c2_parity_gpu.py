"""GPU parity of the BENCHMARKED configuration (BASELINE.json configs[1]: N = 196, bf16 MFMA, default dispatch): the
kernels bench.py times - cluster LSTM forward (time path), row-wave LSTM forward with the input projection fused + 32-sequence BPTT (band path), N-split
BPTT (time path), weight-stationary gate projection, ring NT / TN GEMMs, dual-operand TN weight gradients on the second
stream, the 960-point register FFT - run together through BSRNN_SE / SEModel and are compared with the CPU oracle
(oracle/bsrnn_ref.py) both in its bf16-emulating form (same rounding points) and in plain f32 (the reference
arithmetic), forward + every parameter gradient + one clip / AdamW step.  `ops.launch_counts()` proves that those
kernels, not their small-shape fallbacks, are what was compared.

Shapes: B = 6 x 1 s @ 48 kHz -> T = 101, K = 34, M = B*T*K = 20,604 rows (>= the 8,192 / 16,384-row thresholds of the
bres / ring kernels), time path 204 sequences x 101 steps, band path 606 sequences x 34 steps; the two band-path
thresholds that depend on the NUMBER of sequences (>= 6,144 for the row-wave forward, >= 4,096 for the 32-row BPTT) are lowered
for the test, the band path skips the cluster forward as it does at C2 (12,832 sequences exceed its capacity) and the
grouped ring GEMM accepts 606-row groups (1,024 by default); everything else is the default dispatch.
"""
import pytest
import torch

from oracle import bsrnn_ref, losses_ref
from tests import parity_log

pytestmark = pytest.mark.gpu

N, B, FS, SECONDS = 196, 6, 48000, 1.0
C2_KERNELS = ("stft960", "nt_bres", "nt_ring", "lstm_fwd_clusterx", "lstm_fwd_rwx", "lstm_bwd_nsplit", "lstm_bwd_stream32",
              "tn_dual", "tn_ring_t", "nt_grouped_ring", "tn_grouped")


@pytest.fixture()
def c2_dispatch(monkeypatch):
    from urgent2026_challenge_track1_amd import ops
    monkeypatch.setattr(ops, "WIDE_MIN_SEQ", 1)
    monkeypatch.setattr(ops, "RW_MIN_SEQ", 1)
    monkeypatch.setattr(ops, "BAND_PATH_NO_CLUSTER", True)
    monkeypatch.setenv("URSE_NT_GROUPED_MIN_M", "512")
    monkeypatch.setattr(ops, "BWD_ROWS16", {"f": 2 | 16})
    ops.launch_counts(reset=True)
    return ops


def _models(L, seed=0):
    from urgent2026_challenge_track1_amd.config import Config
    from urgent2026_challenge_track1_amd.d_model import SEModel
    torch.manual_seed(seed)
    ref = bsrnn_ref.BSRNN_SE(N, L)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    model = SEModel(Config(model_configs={"num_channel": N, "num_layer": L}, compute_dtype="bf16"))
    model.se_model.load_state_dict(ref.state_dict())
    return ref, model.cuda()


def _batch(seed=1):
    g = torch.Generator().manual_seed(seed)
    Ls = int(SECONDS * FS)
    clean = 0.3 * torch.randn(B, 1, Ls, generator=g)
    noisy = clean + 0.1 * torch.randn(B, 1, Ls, generator=g)
    lens = torch.full((B,), Ls, dtype=torch.int32)
    lens[1] = Ls - 3000            # one shorter utterance: frame masking through the whole model
    return clean, noisy, lens


def _oracle_step(ref, clean, noisy, lens, emulate):
    """loss, wav, grads of the oracle (no optimizer step)."""
    ref.zero_grad(set_to_none=True)
    wav, _ = ref(noisy[:, 0], lens, FS, emulate)
    loss = losses_ref.mr_l1_loss(clean[:, 0], wav).mean()
    loss.backward()
    return float(loss), wav.detach(), {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("L", [1, 6])
def test_bf16_c2_kernel_set_matches_oracle(lib, c2_dispatch, L):
    ops = c2_dispatch
    ref, model = _models(L)
    clean, noisy, lens = _batch()
    loss_e, wav_e, g_e = _oracle_step(ref, clean, noisy, lens, True)     # same rounding points as the bf16 MFMA path
    loss_f, wav_f, g_f = _oracle_step(ref, clean, noisy, lens, False)    # the reference arithmetic (f32)

    wav = model.se_model(noisy[:, 0].cuda(), lens, FS)[0]
    loss = ops.mr_l1_loss(clean[:, 0].cuda(), wav).mean()
    loss.backward()
    model.se_model.core._flush_deferred_wgrads()
    torch.cuda.synchronize()
    counts = ops.launch_counts()
    missing = [k for k in C2_KERNELS if counts[k] == 0]
    assert not missing, ("kernels of the benchmarked configuration that did not run", missing, counts)
    assert counts["lstm_fwd_stream"] == 0 and counts["nt_128"] == 0, counts

    wav_c = wav.detach().cpu()
    sc = float(wav_f.abs().max())
    e_wav_e = float((wav_c - wav_e).abs().max()) / sc
    e_wav_f = float((wav_c - wav_f).abs().max()) / sc
    e_loss_e = abs(float(loss) - loss_e) / abs(loss_e)
    e_loss_f = abs(float(loss) - loss_f) / abs(loss_f)
    ge = gf = 0.0
    worst = None
    for n, p in model.se_model.named_parameters():
        if n not in g_e:
            assert torch.all(p.grad == 0), n
            continue
        r_e, r_f = _rel(p.grad.cpu(), g_e[n]), _rel(p.grad.cpu(), g_f[n])
        if r_e > ge:
            ge, worst = r_e, n
        gf = max(gf, r_f)
    l2_e, l2_f = _rel(wav_c, wav_e), _rel(wav_c, wav_f)
    print("C2 kernel set, L=%d: wav vs emulated %.2e / f32 %.2e (rel. L2 %.2e / %.2e); loss %.2e / %.2e; worst grad (rel. L2) %.2e (%s) / %.2e"
          % (L, e_wav_e, e_wav_f, l2_e, l2_f, e_loss_e, e_loss_f, ge, worst, gf))
    parity_log.record("bf16_c2_kernel_set_L%d" % L, shape="B6 x 1 s @ 48 kHz, N=196", vs_emulating_oracle=dict(wav_max_over_peak=e_wav_e, wav_rel_l2=l2_e,
                      loss_rel=e_loss_e, worst_grad_rel_l2=ge), vs_f32_oracle=dict(wav_max_over_peak=e_wav_f, wav_rel_l2=l2_f, loss_rel=e_loss_f,
                                                                                    worst_grad_rel_l2=gf))
    # Bounds = 2x what round 2 / 3 observed on the GPU (L=1: 1.4e-3 / 4.0e-3 wav, 6.6e-3 / 1.06e-2 grads; L=6: 2.5e-3 / 4.4e-3,
    # 4.5e-3 / 1.04e-2; rel. L2 of the waveform 1.5e-3 / 4.2e-3 and 2.6e-3 / 4.3e-3; loss <= 8e-6 / 3.1e-5): a regression by 2x fails.  In bf16 the LOSS meets north_star's 1e-3 against the f32
    # reference arithmetic; waveform samples (4e-3 of the peak) and gradients (1e-2) do NOT - that is what 8-bit operand mantissas
    # give, stated in DESIGN.md section 4 and in bench.py's line ("parity"); the f32 mode below meets 1e-3 on everything.
    # against the oracle that rounds where the kernels round: only summation order and the exp / rcp approximations differ
    assert e_wav_e <= 5e-3 and l2_e <= 5.2e-3 and e_loss_e <= 5e-5 and ge <= 1.3e-2, (e_wav_e, l2_e, e_loss_e, ge, worst)
    # against the f32 reference arithmetic: what bf16 operands cost
    assert e_wav_f <= 9e-3 and l2_f <= 8.7e-3 and e_loss_f <= 1e-4 and gf <= 2.1e-2, (e_wav_f, l2_f, e_loss_f, gf)


def test_f32_full_width_matches_oracle_1e3(lib):
    """north_star's tolerance (1e-3 relative on enhanced waveform, spectrum, loss - and here every gradient) at the REAL width and
    depth: N = 196, L = 6, compute_dtype f32 (exact-f32 MFMA), default dispatch, B = 6 x 1 s @ 48 kHz with one shorter
    utterance, against the f32 oracle (whose BandSplit / dual-path loop are pinned to the reference's in-tree twin)."""
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd.config import Config
    from urgent2026_challenge_track1_amd.d_model import SEModel
    torch.manual_seed(11)
    ref = bsrnn_ref.BSRNN_SE(N, 6)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    model = SEModel(Config(model_configs={"num_channel": N, "num_layer": 6}, compute_dtype="f32"))
    model.se_model.load_state_dict(ref.state_dict())
    model = model.cuda()
    clean, noisy, lens = _batch(seed=12)
    ref.zero_grad(set_to_none=True)
    wav_r, spec_r = ref(noisy[:, 0], lens, FS, False)
    loss_r = losses_ref.mr_l1_loss(clean[:, 0], wav_r).mean()
    loss_r.backward()
    wav, spec = model.se_model(noisy[:, 0].cuda(), lens, FS)
    loss = ops.mr_l1_loss(clean[:, 0].cuda(), wav).mean()
    loss.backward()
    model.se_model.core._flush_deferred_wgrads()
    torch.cuda.synchronize()
    ops.poll_kernel_errors(torch.device("cuda", torch.cuda.current_device()), sync=True)
    wav_c, wav_r = wav.detach().cpu(), wav_r.detach()
    e_wav = float((wav_c - wav_r).abs().max() / wav_r.abs().max())
    sp_c, sp_r = torch.view_as_real(spec.detach().cpu()), torch.view_as_real(spec_r.detach())
    e_spec = float((sp_c - sp_r).abs().max() / sp_r.abs().max())
    e_loss = abs(float(loss) - float(loss_r)) / abs(float(loss_r))
    worst, wname = 0.0, None
    refg = dict(ref.named_parameters())
    for n, p in model.se_model.named_parameters():
        gr = refg[n].grad
        if gr is None:
            assert torch.all(p.grad == 0), n
            continue
        r = max(_rel(p.grad.cpu(), gr), float((p.grad.cpu() - gr).abs().max() / (gr.abs().max() + 1e-30)))
        if r > worst:
            worst, wname = r, n
    print("f32 N=196 L=6: wav %.2e (rel. L2 %.2e), spec %.2e, loss %.2e, worst grad %.2e (%s)"
          % (e_wav, _rel(wav_c, wav_r), e_spec, e_loss, worst, wname))
    parity_log.record("f32_full_width_L6", shape="B6 x 1 s @ 48 kHz, N=196, L=6, compute_dtype f32", wav_max_over_peak=e_wav,
                      wav_rel_l2=_rel(wav_c, wav_r), spec=e_spec, loss_rel=e_loss, worst_grad=worst, meets_1e_3=bool(max(e_wav, e_spec, e_loss, worst) <= 1e-3))
    assert e_wav <= 1e-3 and _rel(wav_c, wav_r) <= 1e-3 and e_spec <= 1e-3 and e_loss <= 1e-3 and worst <= 1e-3, \
        (e_wav, e_spec, e_loss, worst, wname)


def test_bf16_c2_train_step_matches_oracle(lib, c2_dispatch):
    """SEModel.training_step + backward (wgrads on the second stream) + clip 0.5 + AdamW at N = 196, L = 2, bf16, against the
    oracle's train step run with the same rounding points."""
    ops = c2_dispatch
    ref, model = _models(2, seed=3)
    (opt,), _ = model.configure_optimizers()
    opt_r = losses_ref.make_optimizer(ref.parameters())
    clean, noisy, lens = _batch(seed=4)
    fs_t = torch.tensor(FS, dtype=torch.int32)
    # oracle step with bf16 emulation
    opt_r.zero_grad(set_to_none=True)
    wav, _ = ref(noisy[:, 0], lens, FS, True)
    loss_r = losses_ref.mr_l1_loss(clean[:, 0], wav).mean()
    loss_r.backward()
    gn_r = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
    before = {n: p.detach().clone() for n, p in ref.named_parameters()}
    opt_r.step()
    loss = model.training_step((clean.cuda(), noisy.cuda(), fs_t, lens))
    loss.backward()
    model.optimizer_step(opt)
    torch.cuda.synchronize()
    ops.poll_kernel_errors(torch.device("cuda", torch.cuda.current_device()), sync=True)
    assert abs(float(loss) - float(loss_r)) <= 1e-3 * abs(float(loss_r))
    assert abs(float(model.logged["Grad_norm"]) - float(gn_r)) <= 2e-2 * float(gn_r)
    # first Adam step = lr * sign(g) (up to eps): compare the UPDATE direction; elements whose gradient is ~0 may flip
    tot = flipped = 0
    for n, p in model.se_model.named_parameters():
        d_m = p.detach().cpu() - before[n]
        d_r = dict(ref.named_parameters())[n].detach() - before[n]
        tot += d_m.numel()
        flipped += int(((d_m - d_r).abs() > 5e-4).sum())
    assert flipped <= 2e-2 * tot, (flipped, tot)
    counts = ops.launch_counts()
    assert counts["tn_dual"] > 0 and counts["lstm_fwd_clusterx"] > 0 and counts["lstm_bwd_stream32"] > 0, counts


def test_both_paths_in_rounds_through_the_model_are_bit_identical_to_one_round(lib, monkeypatch):
    """Round 6: the fused cluster forward takes any number of sequences in ROUNDS (the band path at C2: 12; the time path above 33 utterances per GPU: 2).
    A sequence's arithmetic does not depend on the cluster, round or row it gets, so the model's enhanced waveform must not change by a bit when the SAME
    batch is forced into rounds: a reservation of 200 CUs leaves 3 clusters per direction = 192 slots for the time path's 204 and the band path's 606
    sequences (2 and 4 rounds).  Also: the dispatch really went that way (launch counts; no row-wave, no streaming forward)."""
    from urgent2026_challenge_track1_amd import ops
    L = 2
    ref, model = _models(L)
    model.eval()
    clean, noisy, lens = _batch()
    H, Hp = 2 * N, ops.kpad(ops.pad_to(2 * N, 16), torch.bfloat16)
    monkeypatch.setattr(ops, "band_clusterx_pays", lambda H_, Hp_, n_seq, seq_len=34: (ops.lstm_clusterx_plan(H_, Hp_, n_seq) or [0] * 7)[6] > 1)
    with torch.no_grad():
        ops.launch_counts(reset=True)
        wav0 = model.se_model(noisy[:, 0].cuda(), lens, FS)[0].clone()
        c0 = dict(ops.launch_counts())
        assert ops.lstm_clusterx_plan(H, Hp, B * 34)[6] == 1 and c0["lstm_fwd_clusterx"] == 2 * L, c0
        with ops.reserve_cus(co_resident=200):
            pt, pb = ops.lstm_clusterx_plan(H, Hp, B * 34), ops.lstm_clusterx_plan(H, Hp, B * 101)
            assert pt[1] == 3 and pt[6] == 2 and pb[6] == 4, (pt, pb)
            ops.launch_counts(reset=True)
            wav1 = model.se_model(noisy[:, 0].cuda(), lens, FS)[0].clone()
            c1 = dict(ops.launch_counts())
    ops.poll_kernel_errors(torch.device("cuda", torch.cuda.current_device()), sync=True)
    assert c1["lstm_fwd_clusterx"] == 2 * L and c1["lstm_fwd_rwx"] == 0 and c1["lstm_fwd_stream"] == 0 and c1["lstm_fwd_cluster"] == 0, c1
    assert torch.isfinite(wav1).all() and torch.equal(wav0, wav1)
