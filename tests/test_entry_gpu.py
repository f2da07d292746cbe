"""GPU: the entry points a user of the reference would call - train_se.fit for both model types, dynamic mixing as the
training feed, Lightning-shaped checkpoints incl. checkpoint['ema'], inference.py's SE -> Flow fallback - and the
optimizer semantics that only show with mixed sampling rates or several ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import bsrnn_ref, losses_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(**kw):
    from urgent2026_challenge_track1_amd.config import Config
    base = dict(model_configs={"num_channel": 16, "num_layer": 1}, compute_dtype="f32", seed=7, batch_size=2,
                num_worker=0, train_set_path="synthetic:8", valid_set_path="synthetic", val_check_interval=2,
                num_train_epochs=2, save_top_k=1, resume=False, synthetic_seconds=0.25)
    base.update(kw)
    return Config(**base)


def test_adamw_leaves_unused_bands_alone_like_torch(lib):
    """48 kHz step, then two 16 kHz steps: the bands above 8 kHz get no gradient in steps 2-3; torch.optim.AdamW skips
    such parameters entirely (no decay of p / m / v, no step count) - and so must the fused optimizer."""
    from urgent2026_challenge_track1_amd.d_model import SEModel
    torch.manual_seed(11)
    ref = bsrnn_ref.BSRNN_SE(16, 1)
    opt_r = losses_ref.make_optimizer(ref.parameters())
    model = SEModel(_cfg())
    model.se_model.load_state_dict(ref.state_dict())
    model = model.cuda()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(2)
    hi = "bsrnn.bsrnn.mask_decoder.mlp_mask.33.1.weight"          # band 33 exists only at 44.1 / 48 kHz
    snap = None
    for it, fs in enumerate((48000, 16000, 16000)):
        n = fs // 5
        clean = 0.3 * torch.randn(2, 1, n, generator=g)
        noisy = clean + 0.1 * torch.randn(2, 1, n, generator=g)
        lens = torch.tensor([n, n], dtype=torch.int32)
        losses_ref.train_step(ref, opt_r, clean, noisy, fs, lens)
        loss = model.training_step((clean.cuda(), noisy.cuda(), torch.tensor(fs, dtype=torch.int32), lens))
        loss.backward()
        model.optimizer_step(opt)
        p_hi = dict(model.se_model.named_parameters())[hi].detach().clone()
        if it == 0:
            snap = p_hi
        else:
            assert torch.equal(p_hi, snap), "a band without gradient moved in step %d" % (it + 1)
    steps = opt.slot_steps.cpu().tolist()
    assert steps[0] == 3 and steps[1] == 3 and steps[1 + 33] == 1 and steps[1 + 27] == 1 and steps[1 + 26] == 3, steps
    refp = dict(ref.named_parameters())
    tot = bad = 0
    for n_, p in model.se_model.named_parameters():
        d = (p.detach().cpu() - refp[n_].detach()).abs()
        assert d.max().item() <= 2 * 1e-3 * 3 + 1e-5, n_
        tot += d.numel()
        bad += int((d > 5e-5).sum())
    assert bad <= 2e-3 * tot, (bad, tot)


def test_nan_loss_trains_on_zero_gradients(lib):
    """d_model.py:75-77: a NaN loss skips the step's gradient (here: decided on the device, gradients exactly zero)."""
    from urgent2026_challenge_track1_amd.d_model import SEModel
    model = SEModel(_cfg()).cuda()
    clean = 0.3 * torch.randn(2, 1, 4800)
    noisy = clean.clone()
    clean[0, 0, 100] = float("nan")
    loss = model.training_step((clean.cuda(), noisy.cuda(), torch.tensor(48000, dtype=torch.int32), torch.tensor([4800, 4800])))
    loss.backward()
    model.se_model.core._flush_deferred_wgrads()
    assert torch.isnan(loss) and torch.all(model.se_model.core.flat_grads[:model.se_model.core.flat_params.numel()] == 0)


def _flow_cfg(**kw):
    return _cfg(model_type="flowse", bsrnn_hidden=16, num_layer=1, learning_rate=1e-4, n_fft=1536, hop_length=384,
                spec_abs_exponent=0.667, spec_factor=0.065, sigma_min=0.05, sigma_max=0.5, t_eps=0.03, T_rev=1.0,
                ema_decay=0.999, loss_type="mse", train_name="flow", **kw)


def test_flow_fit_checkpoint_ema_and_inference_fallback(lib, tmp_path):
    """train_se.py:50-53 model select -> FlowSEModel through the same loop; checkpoint['ema'] (flow_model.py:95-96);
    inference.py:30-33 falls back from SEModel to FlowSEModel and eval() swaps the EMA weights in (:99-113)."""
    from urgent2026_challenge_track1_amd import inference, train_se
    from urgent2026_challenge_track1_amd.dataset import SyntheticPairDataset, read_audio, write_audio
    from urgent2026_challenge_track1_amd.flow_model import FlowSEModel
    os.chdir(tmp_path)
    cfg = _flow_cfg(train_tag="t")
    model, steps = train_se.fit(cfg, max_steps=3, log_every=1)
    assert isinstance(model, FlowSEModel) and steps == 3 and model.ema.num_updates == 3
    assert "val_loss" in model.logged and "sisnr" in model.logged       # validation_step ran enhance(N=10) on batch 0
    ck_files = [f for f in os.listdir(train_se.ckpt_dir(cfg)) if "val_loss" in f]
    assert len(ck_files) == 1
    ck = torch.load(os.path.join(train_se.ckpt_dir(cfg), ck_files[0]), map_location="cpu", weights_only=False)
    assert "ema" in ck and ck["ema"]["num_updates"] == 2 and any(k.startswith("dnn.") for k in ck["state_dict"])
    assert len(ck["ema"]["shadow_params"]) == len(list(model.parameters()))
    loaded = inference.load_from_checkpoint(os.path.join(train_se.ckpt_dir(cfg), ck_files[0]))
    assert isinstance(loaded, FlowSEModel)
    raw = loaded.dnn.flat_params.clone()
    loaded.eval()
    assert torch.equal(loaded.dnn.flat_params, loaded.ema.shadow) and not torch.equal(raw, loaded.ema.shadow)
    loaded.train()
    assert torch.equal(loaded.dnn.flat_params, raw)
    loaded.eval()
    ds = SyntheticPairDataset(1, fs_list=(48000,), seconds=0.25)
    _, noisy, fs, L = ds[0]
    write_audio(str(tmp_path / "in.wav"), noisy[0], fs, "FLOAT")
    (tmp_path / "in.scp").write_text("utt1 %s\n" % (tmp_path / "in.wav"))
    args = inference.parser().parse_args(["--input_scp", str(tmp_path / "in.scp"), "--output_dir", str(tmp_path / "out"),
                                          "--ckpt_path", os.path.join(train_se.ckpt_dir(cfg), ck_files[0])])
    inference.main(args)
    y, fs2 = read_audio(str(tmp_path / "out" / "wav" / "utt1.wav"))
    assert fs2 == fs and y.shape[1] == L and abs(abs(y).max() - 0.9) < 1e-3


def test_flow_init_from_and_resume_keep_the_ema_on_the_device(lib, tmp_path):
    """every flow checkpoint carries 'ema'; `init_from` used to create the EMA shadow on the host (before .to(dev)) and
    the first optimizer_step then handed a host pointer to urse_ema_update.  init_from = plain weight load (reference
    train_se.py:55-59), EMA starts from the loaded weights; resume restores checkpoint['ema']; both fine-tune on the GPU."""
    from urgent2026_challenge_track1_amd import train_se
    os.chdir(tmp_path)
    cfg = _flow_cfg(train_tag="a")
    model, _ = train_se.fit(cfg, max_steps=2, log_every=1)
    ck_dir = train_se.ckpt_dir(cfg)
    ck_path = os.path.join(ck_dir, [f for f in os.listdir(ck_dir) if "val_loss" in f][0])
    ck = torch.load(ck_path, map_location="cpu", weights_only=False)
    # warm start of a NEW run from that checkpoint
    cfg2 = _flow_cfg(train_tag="b", init_from=ck_path)
    m2, steps = train_se.fit(cfg2, max_steps=2, log_every=1)
    assert steps == 2 and m2.ema.shadow.is_cuda and m2.ema.num_updates == 2
    assert torch.isfinite(m2.ema.shadow).all() and torch.isfinite(m2.dnn.flat_params).all()
    # resume of the first run: EMA state restored from the checkpoint, on the device, and training continues
    cfg3 = _flow_cfg(train_tag="a", resume=True)
    m3, steps3 = train_se.fit(cfg3, max_steps=ck["global_step"] + 1, log_every=1)
    assert m3.ema.shadow.is_cuda and m3.ema.num_updates == ck["ema"]["num_updates"] + 1 and steps3 == ck["global_step"] + 1
    # an EMA created on the host follows the model to the device
    from urgent2026_challenge_track1_amd.flow_model import FlowSEModel
    m4 = FlowSEModel(cfg)
    m4.init_ema()
    assert not m4.ema.shadow.is_cuda
    m4 = m4.cuda()
    m4.ema.update()
    assert m4.ema.shadow.is_cuda


def _write_source_set(root, fs_list=(16000,), n_speech=6):
    """a tiny dynamic-mixing corpus on disk in the reference's layout (dataset.py:453-460)."""
    from urgent2026_challenge_track1_amd.dataset import SyntheticPairDataset, write_audio
    os.makedirs(root / "wav", exist_ok=True)
    rng = np.random.default_rng(0)
    rows = {k: [] for k in ("speech_sources", "noise_scoures", "rirs", "wind_noise_scoures", "source_length")}
    for fs in fs_list:
        for i in range(n_speech):
            n = int(rng.integers(fs // 3, fs // 2))
            x = SyntheticPairDataset.speech_like(rng, n, fs)
            p = root / "wav" / ("sp%d_%d.wav" % (fs, i))
            write_audio(str(p), x, fs, "FLOAT")
            rows["speech_sources"].append("sp%d_%d %d %s" % (fs, i, fs, p))
            rows["source_length"].append("sp%d_%d %d" % (fs, i, n))
        for i in range(3):
            p = root / "wav" / ("nz%d_%d.wav" % (fs, i))
            write_audio(str(p), 0.1 * rng.standard_normal(int(rng.integers(fs // 4, fs))), fs, "FLOAT")
            rows["noise_scoures"].append("nz%d_%d %d %s" % (fs, i, fs, p))
            p = root / "wav" / ("rir%d_%d.wav" % (fs, i))
            h = rng.standard_normal(fs // 8) * np.exp(-np.arange(fs // 8) / (0.02 * fs))
            write_audio(str(p), h / np.abs(h).max(), fs, "FLOAT")
            rows["rirs"].append("rir%d_%d %d %s" % (fs, i, fs, p))
        p = root / "wav" / ("wn%d.wav" % fs)
        write_audio(str(p), 0.1 * rng.standard_normal(fs // 2), fs, "FLOAT")
        rows["wind_noise_scoures"].append("wind_noise%d %d %s" % (fs, fs, p))
    for k, v in rows.items():
        (root / (k + ".scp")).write_text("\n".join(v) + "\n")


def _write_valid_set(root, fs=16000, n=4):
    from urgent2026_challenge_track1_amd.dataset import SyntheticPairDataset, write_audio
    os.makedirs(root / "wav", exist_ok=True)
    ds = SyntheticPairDataset(n, fs_list=(fs,), seconds=0.3)
    rows = {k: [] for k in ("spk1.scp", "wav.scp", "utt2fs", "speech_length.scp")}
    for i in range(n):
        clean, noisy, _, L = ds[i]
        write_audio(str(root / "wav" / ("c%d.wav" % i)), clean[0], fs, "FLOAT")
        write_audio(str(root / "wav" / ("n%d.wav" % i)), noisy[0], fs, "FLOAT")
        rows["spk1.scp"].append("u%d %s" % (i, root / "wav" / ("c%d.wav" % i)))
        rows["wav.scp"].append("u%d %s" % (i, root / "wav" / ("n%d.wav" % i)))
        rows["utt2fs"].append("u%d %d" % (i, fs))
        rows["speech_length.scp"].append("u%d %d" % (i, L))
    for k, v in rows.items():
        (root / k).write_text("\n".join(v) + "\n")


def test_fit_with_dynamic_mixing_feed(lib, tmp_path):
    """config C3's producer: DynamicMixingDataset draws recipes on the host, the simulator runs on the GPU inside the loop
    (train_se.to_device -> RawMixBatch.materialise -> mixing.simulate_recipes), validation from a pre-simulated set."""
    from urgent2026_challenge_track1_amd import train_se
    _write_source_set(tmp_path / "train")
    _write_valid_set(tmp_path / "valid")
    os.chdir(tmp_path)
    np.random.seed(5)
    cfg = _cfg(train_set_path=str(tmp_path / "train"), valid_set_path=str(tmp_path / "valid"), train_set_dynamic_mixing=True,
               max_duration=8000, train_tag="dm", num_train_epochs=3)
    model, steps = train_se.fit(cfg, max_steps=5, log_every=1)
    assert steps == 5 and torch.isfinite(model.logged["train_loss"])
    assert len([f for f in os.listdir(train_se.ckpt_dir(cfg)) if "val_loss" in f]) == 1


def test_device_prefetcher_equals_inline_materialise(lib, tmp_path):
    """staging a dynamic-mixing batch one step ahead on the side stream gives the tensors the inline path gives, while the main
    stream is busy with other work."""
    import copy
    from urgent2026_challenge_track1_amd import train_se
    from urgent2026_challenge_track1_amd.dataset import DynamicMixingDataset, collate_dynamic
    _write_source_set(tmp_path / "train", n_speech=8)
    root = tmp_path / "train"
    ds = DynamicMixingDataset(*[str(root / (k + ".scp")) for k in ("speech_sources", "noise_scoures", "rirs", "wind_noise_scoures",
                                                                   "source_length")], max_duration=6000)
    np.random.seed(9)
    raws = [collate_dynamic([ds[(4 * k + b) % len(ds)] for b in range(4)]) for k in range(3)]
    dev = torch.device("cuda")
    inline = [copy.deepcopy(r).materialise(dev) for r in raws]
    busy = torch.randn(2048, 2048, device=dev)
    got = []
    for out in train_se.DevicePrefetcher([copy.deepcopy(r) for r in raws], dev):
        for _ in range(20):
            busy = busy @ busy * 1e-3                    # main-stream work the staged batch overlaps with
        got.append(out)
    assert len(got) == 3
    for a, b in zip(inline, got):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and int(a[2]) == int(b[2])


def test_two_rank_fit_mixed_fs_uneven_shards(lib, tmp_path):
    """train_se.fit (not bench.py) with two gloo ranks on one GPU, 16 / 48 kHz utterances of varying length: the ranks see
    different K per step (unused-band zeros through the real model), their shards give different batch counts
    (equalise_batch_counts truncates to the minimum), and after two epochs both hold bit-identical weights."""
    script = tmp_path / "run.py"
    script.write_text(
        "import os, sys, json, torch\n"
        "sys.path.insert(0, %r)\n"
        "from urgent2026_challenge_track1_amd import train_se\n"
        "from urgent2026_challenge_track1_amd.config import Config\n"
        "os.chdir(%r)\n"
        "cfg = Config(model_configs={'num_channel': 16, 'num_layer': 1}, compute_dtype='bf16', seed=3, batch_size=2, num_worker=0,\n"
        "             train_set_path='synthetic:22', valid_set_path='synthetic', val_check_interval=100000, num_train_epochs=2,\n"
        "             resume=False, synthetic_seconds=0.2, synthetic_fs=(16000, 48000), train_tag='r' + os.environ['RANK'])\n"
        "model, steps = train_se.fit(cfg)\n"
        "core = model.se_model.core\n"
        "torch.cuda.synchronize()\n"
        "json.dump({'steps': steps, 'sum': float(core.flat_params.double().sum()), 'abs': float(core.flat_params.double().abs().sum())},\n"
        "          open('rank%%s.json' %% os.environ['RANK'], 'w'))\n" % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", URSE_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2))
    assert a["steps"] == b["steps"] and a["steps"] > 0
    assert a["sum"] == b["sum"] and a["abs"] == b["abs"], (a, b)
