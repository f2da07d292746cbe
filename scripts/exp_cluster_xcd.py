"""Time path forward (1,088 sequences x 401 steps per direction): static clusters (consecutive blockIdx = seven XCDs) against clusters formed
from workgroups that read the same XCC id; bit equality of the outputs and time per launch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urgent2026_challenge_track1_amd import ops
dev = "cuda"
N, B, T, K = 196, 32, 401, 34
H, Hp = 2 * N, 416
torch.manual_seed(0)
whh = torch.randn(2 * 4 * H, H, device=dev) * 0.05
whhq = torch.empty(2 * ((H + 3) // 4) * (Hp // 32) * 512, device=dev, dtype=torch.bfloat16)
ops.call("lstm_pack_quads", whh, whhq, H, Hp, ops.BF16, ops.stream_ptr())
M = B * T * K
sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
res = {}
for xa in (False, True):
    g = gx.clone()
    h, c, err = ops.lstm_fwd_cluster(g, whhq, H, Hp, xcd_aware=xa, **sm)
    torch.cuda.synchronize()
    assert int(err.item()) == 0, "kernel error flag"
    res[xa] = (g, h, c)
eq = lambda a, b: bool(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b))
print("xcd-aware == static: gates %s  h %s  c %s" % tuple(eq(res[True][i], res[False][i]) for i in range(3)), flush=True)
key = [k for k in ops._cluster_ws if k[0] == torch.device(dev, 0) or True][0]
cnt = ops._cluster_ws[key][1].cpu().tolist()
print("registrations per XCC id:", cnt[:8], "total", cnt[8], flush=True)
del res
g = gx.clone()
for xa in (False, True, False, True):
    ts = []
    for _ in range(4):
        g.copy_(gx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.lstm_fwd_cluster(g, whhq, H, Hp, xcd_aware=xa, **sm)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("xcd_aware %-5s %.3f ms (min of 4: %s)" % (xa, min(ts), " ".join("%.3f" % v for v in ts)), flush=True)
# helper waves (URSE_CLUSTER_HELPERS: 0 = 14 working waves do everything, 2 = two helper waves own the plain stores and the pre-activations)
outs = {}
for hp in ("0", "2"):
    os.environ["URSE_CLUSTER_HELPERS"] = hp
    g = gx.clone()
    h, c, err = ops.lstm_fwd_cluster(g, whhq, H, Hp, xcd_aware=True, **sm)
    torch.cuda.synchronize()
    assert int(err.item()) == 0, "kernel error flag"
    outs[hp] = (g, h.clone(), c.clone())
print("helper waves == none: gates %s  h %s  c %s" % tuple(eq(outs["0"][i], outs["2"][i]) for i in range(3)), flush=True)
del outs
g = gx.clone()
for hp in ("0", "2", "0", "2"):
    os.environ["URSE_CLUSTER_HELPERS"] = hp
    ts = []
    for _ in range(4):
        g.copy_(gx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.lstm_fwd_cluster(g, whhq, H, Hp, xcd_aware=True, **sm)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("helpers %s  %.3f ms (min of 4: %s)" % (hp, min(ts), " ".join("%.3f" % v for v in ts)), flush=True)
