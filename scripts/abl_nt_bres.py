"""Weight-stationary NT kernel (URSE_NT_BRES=1) against the ring kernel on the gate projection and ragged relatives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
dev, bf = "cuda", torch.bfloat16
def bench(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, M, N, K, act in (("ih fwd", 32 * 401 * 34, 3136, 224, 0), ("ragged", 32 * 401 * 34 - 77, 3000, 160, 1), ("small M", 9000, 1800, 96, 0), ("dgrad fc", 32 * 401 * 34, 800, 224, 0), ("N=448", 32 * 401 * 34, 448, 224, 0)):
    a = (torch.randn(M, K, device=dev) * 0.1).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    b = torch.randn(N, device=dev)
    out = {}
    for mode in ("0", "1"):
        os.environ["URSE_NT_BRES"] = mode
        f = lambda: ops.gemm_nt(a, w, bias=b, act=act, out_dtype=bf)
        c = f()
        out[mode] = (bench(f), c)
    ref = (a[:2048].float() @ w.float().t()) + b
    if act: ref = torch.tanh(ref)
    e1 = (out["1"][1][:2048].float() - ref).abs().max().item()
    print("%-8s M=%d N=%d K=%d: ring %.3f ms | weight-stationary %.3f ms   err %.3g  bit-equal %s" %
          (name, M, N, K, out["0"][0], out["1"][0], e1, torch.equal(out["0"][1], out["1"][1])), flush=True)
