"""Fused row-wave forward (input projection inside the recurrence, csrc/lstm_rwx.hip) against the two-kernel form {gate GEMM, lstm_rw}:
agreement of h / c / gates, accuracy of both against torch.nn.LSTM in f32 on a small shape, time per launch at the C2 band-path shape."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urgent2026_challenge_track1_amd import ops
dev = "cuda"
N, H = 196, 392
dt = torch.bfloat16
torch.manual_seed(0)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
Hp, Np = pk["Hp"], pk["Np"]
assert "wx" in pk

def two_kernel(xr, sm, save=True):
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    h, c = ops.lstm_fwd_rw(gx, pk["whhb"], H, Hp, save=save, **sm)
    return gx, h, c

def fused(xr, sm, save=True):
    return ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, Hp, save=save, **sm)

# ---- accuracy against nn.LSTM (f32, CPU) on 300 sequences x 34 steps, plus a ragged / strided map
for name, ns, sl, mapf in (("band 300 x 34", 300, 34, lambda ns, sl: dict(n_seq=ns, seq_len=sl, inner=1, outer=sl, stride=1)),
                           ("band 37 x 5", 37, 5, lambda ns, sl: dict(n_seq=ns, seq_len=sl, inner=1, outer=sl, stride=1))):
    sm = mapf(ns, sl)
    x = torch.randn(ns, sl, N)
    with torch.no_grad():
        y, _ = lstm(x)
    xr = ops.pack2d(x.reshape(ns * sl, N).to(dev), ns * sl, Np, dt)
    a = two_kernel(xr, sm)
    b = fused(xr, sm)
    torch.cuda.synchronize()
    yr = y.reshape(ns * sl, 2 * H)
    ea = (a[1][:, :2 * H].float().cpu() - yr).abs()
    eb = (b[1][:, :2 * H].float().cpu() - yr).abs()
    d = (a[1].float() - b[1].float()).abs()
    dg = (a[0].float() - b[0].float()).abs()
    dc = (a[2] - b[2]).abs()
    print("%-14s |h - nn.LSTM|: two-kernel max %.2e mean %.2e ; fused max %.2e mean %.2e ; fused vs two-kernel: h max %.2e mean %.2e, gates max %.2e, c max %.2e"
          % (name, ea.max(), ea.mean(), eb.max(), eb.mean(), d.max(), d.mean(), dg.max(), dc.max()), flush=True)
    b0 = fused(xr, sm, save=False)
    print("   save=0: h equal to save=1: %s" % bool(torch.equal(b0[1].view(torch.int16), b[1].view(torch.int16))), flush=True)

# ---- C2 band path: time
B, T, K = 32, 401, 34
M = B * T * K
sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
xr = ops.pack2d(torch.randn(M, N, device=dev), M, Np, dt)
a = two_kernel(xr, sm); b = fused(xr, sm); torch.cuda.synchronize()
d = (a[1].float() - b[1].float()).abs()
print("C2 band path: fused vs two-kernel h max %.2e mean %.2e; gates max %.2e" % (d.max(), d.mean(), (a[0].float() - b[0].float()).abs().max()), flush=True)
del a, b, d
for kind in ("gemm", "rw", "two", "fused", "two", "fused"):
    ts = []
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if kind == "gemm":
            ops.gemm_nt(xr, pk["wih"], pk["bias"])
        elif kind == "rw":
            ops.lstm_fwd_rw(gx, pk["whhb"], H, Hp, save=True, **sm)
        elif kind == "two":
            two_kernel(xr, sm)
        else:
            fused(xr, sm)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-6s %.3f ms (min of 4: %s)" % (kind, min(ts), " ".join("%.3f" % v for v in ts)), flush=True)
