"""Full-size run of the flow model (config C4: n_fft 1536 / hop 384, N = 384, 6 layers, F 769, K 48; B = 2 x 4 s @ 48 kHz):
one training step (forward_step + backward + clip + AdamW + EMA) and an Euler enhancement, timed.  Diagnostic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd.config import Config
from urgent2026_challenge_track1_amd.flow_model import FlowSEModel
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = FlowSEModel(Config(bsrnn_hidden=384, num_layer=6, compute_dtype=dt, sigma_min=0.05, sigma_max=0.5)).cuda()
(opt,), _ = m.configure_optimizers()
B, L = 2, 192000
g = torch.Generator().manual_seed(0)
clean = (0.2 * torch.randn(B, 1, L, generator=g)).cuda()
noisy = clean + (0.05 * torch.randn(B, 1, L, generator=g)).cuda()
batch = (clean, noisy, torch.tensor(48000, dtype=torch.int32), torch.tensor([L] * B))
def step():
    loss = m.training_step(batch)
    loss.backward()
    m.optimizer_step(opt)
    return loss
loss = step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2): loss = step()
torch.cuda.synchronize()
print("flow train step (B=2 x 4 s, %s): %.1f ms/step, loss %.4f, peak %.1f GB" % (dt, (time.perf_counter() - t0) / 2 * 1e3, float(loss),
      torch.cuda.max_memory_allocated() / 1e9), flush=True)
with torch.no_grad():
    y = noisy[:1, 0]
    out = m.enhance(y, 48000, torch.tensor([L]), N=15); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.enhance(y, 48000, torch.tensor([L]), N=15); torch.cuda.synchronize()
print("enhance (1 x 4 s, Euler N=15): %.1f ms, finite %s, shape %s" % ((time.perf_counter() - t0) * 1e3, bool(torch.isfinite(out).all()),
      tuple(out.shape)), flush=True)
