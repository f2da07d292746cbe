"""PESQ kernel throughput vs pairs per launch (4 s @ 16 kHz wide-band pairs resident in HBM)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import metrics
import bench
dev = torch.device("cuda")
fs, L = 16000, 64000
NMAX = int(os.environ.get("PESQ_NMAX", "2048"))
clean, noisy = bench.synth_batch(NMAX, L, fs, 1, dev)
metrics.pesq_batch(clean[:64], noisy[:64], fs); torch.cuda.synchronize()
for n in [v for v in (64, 256, 768, 1024, 2048, 4096, 8192) if v <= NMAX]:
    metrics.pesq_batch(clean[:n], noisy[:n], fs, max_pairs_per_launch=n); torch.cuda.synchronize()   # (allocates the workspace of this size)
    t0 = time.perf_counter()
    m = metrics.pesq_batch(clean[:n], noisy[:n], fs, max_pairs_per_launch=n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pairs/launch %5d: %.1f ms  -> %.0f pairs/s   mean MOS %.3f" % (n, dt * 1e3, n / dt, float(torch.nanmean(m))), flush=True)

m, raw, tr = metrics.pesq_batch(clean[:256], noisy[:256], fs, return_trace=True)
tr = tr.cpu().numpy()
names = ("level filters", "input filter", "DC + alignment IIR", "VAD", "alignment + splitting", "perceptual")
vals = [(tr[:, 5 + i // 2] >> (16 * (i % 2)) & 0xffff) * 0.064 for i in range(6)]
print("per-pair stage times, ms (mean / max over 256 pairs): " + " | ".join("%s %.1f / %.1f" % (n, v.mean(), v.max()) for n, v in zip(names, vals))
      + " | utterances mean %.1f" % tr[:, 1].mean())
