"""Dual weight-gradient GEMM (gemm_tn_dual224_kernel) at the C2 shapes: two stages per barrier with the DMA queue drained (shipped, depth 2)
against three stages in flight with one barrier per stage (URSE_TN224_DEPTH=3).  One process, interleaved rounds (the switch is read per call)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
ALT = int(os.environ.get("EXP_ALT_DEPTH", "4"))      # 4: DMA issue interleaved with the MFMA groups (round 6); 3: three stages in flight (round 5, needs -DURSE_EXPERIMENTS)
M, N, H = 32 * 401 * 34, 196, 392
dev, bf = "cuda", torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(bf)
dg, xn, hout = r(M, 8 * H), r(M, 224), r(M, 800)
xn[:, N:] = 0
hout[:, 2 * H:] = 0
flops = 2.0 * M * 4 * H * (196 + H)
def run(depth, wgs, inner, period):
    os.environ["URSE_TN224_DEPTH"] = str(depth)
    gw1, gw2, cs = torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, H, device=dev), torch.zeros(4 * H, device=dev)
    f = lambda: ops.gemm_tn_dual(dg[:, :4 * H], xn, gw1, cs, hout[:, :H], gw2, 4 * H, N, H, -inner, inner, period, 0, perm_h=H, target_wgs=wgs)
    f(); torch.cuda.synchronize()
    gw1.zero_(); gw2.zero_(); cs.zero_()
    f(); torch.cuda.synchronize()
    res = (gw1.clone(), gw2.clone(), cs.clone())
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), res
for (inner, period, name) in ((34, 401, "time path"), (1, 34, "band path")):
    for wgs in (256, 112, 84):
        out = {}
        for rnd in range(3):
            for depth in (2, ALT):
                ms, res = run(depth, wgs, inner, period)
                out.setdefault(depth, []).append(ms)
                out["res%d" % depth] = res
        d = max((a - b).abs().max().item() / (b.abs().max().item() + 1e-30) for a, b in zip(out["res2"], out["res%d" % ALT]))
        print("%s, %3d workgroups: depth 2 %s ms (%.0f TF/s) | depth %d %s ms (%.0f TF/s) | max rel. diff %.1e"
              % (name, wgs, " ".join("%.3f" % v for v in out[2]), flops / min(out[2]) / 1e9, ALT, " ".join("%.3f" % v for v in out[ALT]),
                 flops / min(out[ALT]) / 1e9, d), flush=True)
