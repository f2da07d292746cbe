"""The dual weight-gradient GEMM at the C2 shapes: bf16 x bf16 against bf16 gradients x f16 activations (converted in registers), alone, at the
workgroup counts the train step gives it (256 = whole chip, 112 / 84 = beside the BPTTs).  python scripts/exp_tn_mixed.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
H, N, R = 392, 196, 32 * 401 * 34
g = torch.Generator().manual_seed(0)
A = (0.1 * torch.randn(R, 4 * H, generator=g)).bfloat16().cuda()
X16 = torch.zeros(R, 224, dtype=torch.float16); X16[:, :N] = torch.randn(R, N, generator=g).half(); X16 = X16.cuda()
H16 = torch.zeros(R, 416, dtype=torch.float16); H16[:, :H] = torch.tanh(torch.randn(R, H, generator=g)).half(); H16 = H16.cuda()
Xb, Hb = X16.float().bfloat16(), H16.float().bfloat16()
D = (0.1 * torch.randn(R, 224, generator=g)).bfloat16().cuda()
HH16 = torch.zeros(R, 800, dtype=torch.float16, device="cuda"); HH16[:, :784] = torch.tanh(torch.randn(R, 784, device="cuda")).half()
HHb = HH16.float().bfloat16()
c1, c2, cs = torch.zeros(4 * H, N, device="cuda"), torch.zeros(4 * H, H, device="cuda"), torch.zeros(4 * H, device="cuda")
cf, csf = torch.zeros(N, 784, device="cuda"), torch.zeros(N, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for inner, period, tag in ((34, 401, "time path"), (1, 34, "band path")):
    for wgs in (256, 112, 84):
        r = {}
        for name, X, Hh in (("bf16", Xb, Hb), ("mixed", X16, H16)):
            r[name] = t(lambda: ops.gemm_tn_dual(A, X, c1, cs, Hh, c2, 4 * H, N, H, -inner, inner, period, 0, perm_h=H, target_wgs=wgs))
        print("dual wgrad %-9s target %3d workgroups: bf16 %.3f ms, bf16 x f16 %.3f ms (%+.1f %%)" % (tag, wgs, r["bf16"], r["mixed"], 100 * (r["mixed"] / r["bf16"] - 1)), flush=True)
for wgs in (256, 112):
    r = {}
    for name, Hh in (("bf16", HHb), ("mixed", HH16)):
        r[name] = t(lambda: ops.gemm_tn(D, Hh, cf, colsum=csf, Mo=N, No=784, target_wgs=wgs))
    print("fc wgrad target %3d workgroups: bf16 %.3f ms, bf16 x f16 %.3f ms (%+.1f %%)" % (wgs, r["bf16"], r["mixed"], 100 * (r["mixed"] / r["bf16"] - 1)), flush=True)
