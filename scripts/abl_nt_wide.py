"""A/B of the NT GEMM tile shapes (256x224 vs 128x448) on the step's shapes; getenv is read per call, so one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
dev, bf = "cuda", torch.bfloat16
H = 392
shapes = [("ih fwd time", 32 * 401 * 34, 8 * H, 224, bf, True),
          ("ih fwd band", 32 * 401 * 34, 8 * H, 224, bf, True),
          ("dgrad fc", 32 * 401 * 34, 800, 224, bf, False),
          ("N=896 K=512", 32 * 401 * 34, 896, 512, bf, False),
          ("N=3136 K=800", 32 * 401 * 8, 3136, 800, bf, False),
          ("ih f32out", 32 * 401 * 8, 8 * H, 224, torch.float32, True)]
def bench(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, M, N, K, od, hb in shapes:
    a = (torch.randn(M, K, device=dev) * 0.1).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    b = torch.randn(N, device=dev) if hb else None
    out = {}
    for mode in ("0", "1"):
        os.environ["URSE_NT_WIDE"] = mode
        f = lambda: ops.gemm_nt(a, w, bias=b, out_dtype=od)
        c = f()
        out[mode] = (bench(f), c)
    ref = (a[:4096].float() @ w.float().t()) + (b if hb else 0)
    e0 = (out["0"][1][:4096].float() - ref).abs().max().item()
    e1 = (out["1"][1][:4096].float() - ref).abs().max().item()
    same = torch.equal(out["0"][1], out["1"][1])
    print("%-12s M=%d N=%d K=%d  256x224 %.3f ms | 128x448 %.3f ms  err %.3g / %.3g  bit-equal %s" %
          (name, M, N, K, out["0"][0], out["1"][0], e0, e1, same), flush=True)
