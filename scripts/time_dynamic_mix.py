"""Where the --dynamic-mix step goes: host production, pinning, device simulator (host time vs wall time), train step."""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from urgent2026_challenge_track1_amd.config import Config
from urgent2026_challenge_track1_amd.d_model import SEModel
from urgent2026_challenge_track1_amd.dataset import DynamicMixingDataset, collate_dynamic

dev = torch.device("cuda", 0)
fs, B, seconds = 48000, 32, 4.0
L = int(fs * seconds)
cfg = Config(compute_dtype="bf16", model_configs={"num_channel": 196, "num_layer": 6}, seed=2024)
torch.manual_seed(2024)
model = SEModel(cfg).to(dev)
(opt,), _ = model.configure_optimizers()
src = bench._InMemorySources(tempfile.mkdtemp(prefix="urse_dm_"), fs, seconds, 4 * B, 2024)
ds = DynamicMixingDataset(src.paths["speech_sources"], src.paths["noise_scoures"], src.paths["rirs"], src.paths["wind_noise_scoures"],
                          src.paths["source_length"], max_duration=L, reader=src.read, frames=src.frames)
np.random.seed(2024)


def wall(f, n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return r, th / n * 1e3, (time.perf_counter() - t0) / n * 1e3


def step(batch):
    loss = model.training_step(batch)
    loss.backward()
    model.optimizer_step(opt, None)
    return loss


raws = []
t0 = time.perf_counter()
for k in range(6):
    raws.append(collate_dynamic([ds[(k * B + b) % len(ds)] for b in range(B)]))
print("host: draw + read + stack one batch %.1f ms" % ((time.perf_counter() - t0) / 6 * 1e3))
t0 = time.perf_counter()
for r in raws:
    r.pin_memory()
print("host: pin one batch %.1f ms" % ((time.perf_counter() - t0) / 6 * 1e3))
skipped = {}
batch = raws[0].materialise(dev, skipped)
for _ in range(2):
    step(batch)
_, h, w = wall(lambda: step(batch), 3)
print("train step on a resident batch: host %.1f ms, wall %.1f ms" % (h, w))
for r in raws[1:3]:
    r.materialise(dev, skipped)
res = [wall(lambda r=r: r.materialise(dev, skipped)) for r in raws[3:6]]
print("device simulator alone: host %s ms, wall %s ms" % (["%.1f" % a[1] for a in res], ["%.1f" % a[2] for a in res]))
i = [0]


def both():
    i[0] += 1
    return step(raws[i[0] % 6].materialise(dev, skipped))


both()
_, h, w = wall(both, 4)
print("simulator + train step: host %.1f ms, wall %.1f ms" % (h, w))
print("reverberated utterances per batch:", [sum(1 for n in r.rir_lens if n > 0) for r in raws], "noise width", [r.noise.shape[1] for r in raws])
