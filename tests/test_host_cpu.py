"""CPU: host-side logic -- Config contract vs the reference's own config.py (golden), flat parameter layout,
gradient bucket reducer over gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_matches_reference_behaviour(tmp_path):
    """yaml values override CLI/kwargs and train_tag := yaml basename (reference config.py:41-52), checked against
    values produced by the reference's own Config (tests/golden/ref_config.npz)."""
    from urgent2026_challenge_track1_amd.config import Config, config_parser
    g = np.load(os.path.join(GOLD, "ref_config.npz"))
    yml = tmp_path / "BSRNN_baseline.yaml"
    yml.write_text(
        "train_set_path : data/train_simulation\ntrain_set_dynamic_mixing : False\nvalid_set_path : data/validation\n"
        "val_check_interval : 5000\nmax_duration: 96000\nuse_high_pass : True\nbatch_size: 4\nnum_worker : 2\n"
        "device : \"cuda\"\nnum_gpu : 1\nlearning_rate: 1.e-3\nlr_step_size : 1\nlr_gamma : 0.85\ngradient_clip : 0.5\n"
        "weight_decay: 1.e-6\nadam_epsilon : 1.e-8\nnum_train_epochs : 80\ntrain_version : 0\ntrain_name : 'baseline'\n"
        "save_top_k : 5\nresume : True\nseed : 2024\nmodel_type: \"discriminative\"\ninit_from : 'none'\nse_model: bsrnn\n"
        "model_configs:\n  num_channel: 196\n  num_layer: 6\n")
    c = Config(learning_rate=5e-4, batch_size=7, config_file=str(yml))
    c.read_yaml()
    ref = dict(zip(g["keys"].tolist(), g["values"].tolist()))
    for k, v in ref.items():
        if k == "config_file":
            continue
        assert str(getattr(c, k)) == v, (k, getattr(c, k), v)
    assert str(sorted(c.model_configs.items())) == str(g["model_configs"])
    assert set(g["defaults_keys"].tolist()) <= set(vars(Config()).keys())
    a = config_parser(["--batch_size", "3", "--resume", "false"])
    assert a.batch_size == 3 and a.resume is False and a.learning_rate == 1e-3


def test_flat_layout_and_grad_groups():
    from urgent2026_challenge_track1_amd.bsrnn import BSRNN_SE
    m = BSRNN_SE(16, 2)
    core = m.core
    flat = core.flat_params
    assert flat.numel() >= sum(p.numel() for p in m.parameters())
    for p in m.parameters():
        assert p.data_ptr() >= flat.data_ptr() and p.grad is not None and p.grad.shape == p.shape
    # views alias the flat buffer
    with torch.no_grad():
        core.rnn_time[1].weight_hh_l0_reverse.fill_(3.0)
    o = core._off["l1t.whh"] + 4 * core.H * core.H
    assert torch.all(flat[o:o + 4 * core.H * core.H] == 3.0)
    groups = core.grad_groups()
    assert [g[0] for g in groups] == ["md", "l1f", "l1t", "l0f", "l0t", "bs"]
    covered = sum(g[2] for g in groups)
    assert covered >= sum(p.numel() for p in m.parameters())
    # state_dict round trip keeps aliasing
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert core.rnn_time[1].weight_hh_l0_reverse.data_ptr() >= flat.data_ptr()


class _FakeCore:
    def __init__(self, n):
        self._g = torch.zeros(n)
        self.grad_ready_hook = None

    @property
    def flat_grads(self):
        return self._g

    def grad_groups(self):
        return [("md", 600, 400), ("l0f", 300, 300), ("l0t", 100, 200), ("bs", 0, 100)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from urgent2026_challenge_track1_amd.ddp import GradBucketReducer
    core = _FakeCore(1000)
    red = GradBucketReducer(core, bucket_bytes=1600)      # -> buckets {md}, {l0f,l0t}, {bs}
    assert len(red.buckets) == 3
    for it in range(2):
        core._g.copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1) + it)
        if rank == 1:
            core._g[0:100] = 0.0                            # "unused parameters" on this rank: zeros
        for tag in ("md", "l0f", "l0t", "bs"):
            core.grad_ready_hook(tag)
        scale = red.finish()
        exp = torch.arange(1000, dtype=torch.float32) * 3 + 2 * it
        exp[0:100] = torch.arange(100, dtype=torch.float32) * 1 + it
        assert scale == 0.5 and torch.allclose(core._g, exp), (rank, it)
    q.put(rank)
    dist.destroy_process_group()


def test_grad_bucket_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def test_sampler_shard_rule_and_collate():
    """GroupedBatchSampler: per-fs groups, length-sorted, indices[rank::world] (reference dataset.py:351-366);
    collate_fn right-pads and returns (clean[B,1,T], noisy[B,1,T], fs int32 0-d, lengths int32[B])."""
    from urgent2026_challenge_track1_amd.dataset import GroupedBatchSampler, SyntheticPairDataset, collate_fn
    ds = SyntheticPairDataset(40, fs_list=(16000, 48000), seconds=0.05, vary_length=True)
    lens, srs = ds.get_source_length(), ds.get_srs()
    seen = []
    for rank in range(2):
        s = GroupedBatchSampler(ds, 4, rank, 2, drop_last=False)
        exp = []
        for sr in (16000, 48000):
            idx = sorted([i for i in range(40) if srs[i] == sr], key=lambda i: lens[i])[rank::2]
            exp += idx
        got = [i for b in s for i in b]
        assert sorted(got) == sorted(exp)
        for b in s:
            assert len({srs[i] for i in b}) == 1          # one sampling rate per batch
        seen += got
    assert sorted(seen) == list(range(40))               # ranks partition the data
    batch = collate_fn([ds[0], ds[2]])
    assert batch[0].shape == batch[1].shape and batch[0].shape[1] == 1
    assert batch[0].shape[2] == max(lens[0], lens[2]) and batch[2].dtype == torch.int32 and batch[2].dim() == 0
    assert batch[3].tolist() == [lens[0], lens[2]]
    short = 0 if lens[0] < lens[2] else 1
    assert torch.all(batch[0][short, 0, min(lens[0], lens[2]):] == 0)


def test_wav_io_roundtrip(tmp_path):
    from urgent2026_challenge_track1_amd.dataset import read_audio, write_audio
    x = np.clip(np.random.default_rng(0).standard_normal(1000) * 0.3, -0.99, 0.99).astype(np.float32)
    write_audio(str(tmp_path / "a.wav"), x, 16000, "FLOAT")
    y, fs = read_audio(str(tmp_path / "a.wav"))
    assert fs == 16000 and np.array_equal(y[0], x)
    write_audio(str(tmp_path / "b.wav"), x, 48000)
    y, fs = read_audio(str(tmp_path / "b.wav"))
    assert fs == 48000 and np.abs(y[0] - x).max() <= 1.0 / 32768


def test_device_prefetcher_order_and_end():
    """the one-batch-ahead iterator hands out every batch once, in order, and ends with the loader (host path: no side stream)."""
    import torch
    from urgent2026_challenge_track1_amd.train_se import DevicePrefetcher
    batches = [(torch.full((2, 1, 8), float(i)), torch.full((2, 1, 8), -float(i)), torch.tensor(16000), torch.tensor([8, 8]))
               for i in range(5)]
    got = list(DevicePrefetcher(batches, "cpu"))
    assert len(got) == 5 and all(float(g[0][0, 0, 0]) == i and float(g[1][0, 0, 0]) == -i for i, g in enumerate(got))
    assert list(DevicePrefetcher([], "cpu")) == []
    assert len(list(DevicePrefetcher(batches[:1], "cpu"))) == 1


def test_bench_gpus_n_launches_n_ranks_without_torchrun():
    """bench.py --gpus N outside torch.distributed.run starts N ranks itself, relays rank 0's JSON line and propagates a failing
    rank as a non-zero exit (the rendezvous / relay part runs here on gloo without a GPU; the real step under -m gpu:
    tests/test_train_gpu.py::test_bench_gpus_2_starts_its_own_ranks)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest", "ok"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["sum"] == 3.0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest", "fail"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "error" in json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    # one rank per GPU over RCCL: asking for more ranks than visible GPUs is refused before anything is started
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 64


def test_bench_weight_stream_bound_arithmetic():
    """the supplementary bound of the bench line (recurrent weights streamed from L2 once per step and workgroup): bytes per launch,
    workgroup counts of the two BPTT geometries, the per-CU figure against 34.5 TB/s / 256; forward kernels have none."""
    import bench
    B, T, K, H = 32, 401, 34, 392
    t = bench.l2_port_roofline("lstm_bwd_time", 7.0, B, T, K, H)
    assert t["workgroups"] == 2 * (B * K // 16) == 136 and t["cus"] == 136
    assert t["bytes_per_launch"] == 136 * (T - 1) * 4 * H * H * 2
    assert abs(t["GBs_per_cu"] - t["bytes_per_launch"] / 136 / 7.0e-3 / 1e9) < 1e-9 and 0.4 < t["frac"] < 0.6
    b = bench.l2_port_roofline("lstm_bwd_band", 4.3, B, T, K, H)
    assert b["workgroups"] == 2 * -(-B * T // 32) == 802 and b["cus"] == 256 and b["frac"] < t["frac"]
    assert bench.l2_port_roofline("lstm_fwd_time", 3.5, B, T, K, H) is None


def test_flow_split_bptt_plan_does_not_depend_on_the_nsplit_shadow_size(monkeypatch):
    """VERDICT r4 item 5 (the 87.8 -> 104.3 ms regression of the flow train step): the second queue's sizing beside the N-split BPTT
    (TN_SHADOW_WGS_NSPLIT) must not reach the flow model, whose cooperative split BPTT is planned on the CUs the second queue leaves.  The
    plan queries need no GPU (256 CUs assumed)."""
    import torch
    from urgent2026_challenge_track1_amd import _lib, ops
    _lib.load()
    sm_flow = dict(n_seq=96, seq_len=501, inner=48, outer=501 * 48, stride=48)           # C4 time path: B 2 x 48 bands, 501 frames
    sm_c2 = dict(n_seq=1088, seq_len=401, inner=34, outer=401 * 34, stride=34)
    plans = set()
    for v in (84, 98, 112, 140):
        monkeypatch.setattr(ops, "TN_SHADOW_WGS_NSPLIT", v)
        assert ops.use_nsplit_bwd(768, 768, torch.bfloat16, "t", sm_flow) is False         # H = 768 has no N-split kernel
        target = ops.wgrad_shadow_wgs("t", False)
        assert target == ops.TN_SHADOW_WGS
        with ops.reserve_cus(co_resident=target):
            plans.add(tuple(ops.lstm_split_plan(768, sm_flow["n_seq"])))
        assert ops.use_nsplit_bwd(392, 416, torch.bfloat16, "t", sm_c2) is True and ops.wgrad_shadow_wgs("t", True) == v
        assert ops.wgrad_shadow_wgs("f", True) == ops.TN_SHADOW_WGS_BAND
    assert len(plans) == 1 and next(iter(plans))[0] == 12, plans                             # the 12-way split of the benchmarked flow step


def test_no_cooperative_plan_when_ranks_share_a_gpu(monkeypatch):
    """two processes on one GPU (the gloo test path) must not each bring a cooperative grid - clusters that spin until every member
    has arrived cannot be co-resident with another process's.  With ops.SHARED_GPU_RANKS > 1 every cooperative plan (cluster, cluster2, split,
    N-split) is refused and the streaming kernels run; alone on the device the C2 / C4 plans exist.  The plan queries need no GPU."""
    from urgent2026_challenge_track1_amd import _lib, ops
    _lib.load()
    assert ops.lstm_cluster_plan(392, 416, 1088) is not None and ops.lstm_nsplit_plan(392, 1088) is not None
    assert ops.lstm_cluster2_plan(768, 768, 48) is not None and ops.lstm_split_plan(768, 96) is not None
    monkeypatch.setattr(ops, "SHARED_GPU_RANKS", 2)
    assert ops.lstm_cluster_plan(392, 416, 1088) is None and ops.lstm_nsplit_plan(392, 1088) is None
    assert ops.lstm_cluster2_plan(768, 768, 48) is None and ops.lstm_split_plan(768, 96) is None
