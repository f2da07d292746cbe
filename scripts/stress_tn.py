"""Race screen for the TN ring kernel's two-stages-per-barrier loop: the same dual wgrad 40 times per shape / workgroup
target against an f64-accumulated reference of the first run's inputs; a stale LDS read would show as an outlier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
from urgent2026_challenge_track1_amd._lib import call
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(1)
worst = 0.0
for (B, T, K, H, N) in ((32, 401, 34, 392, 196), (4, 401, 34, 392, 196), (3, 97, 34, 392, 196)):
    M = B * T * K
    dg = (torch.randn(M, 4 * H, device=dev) * 0.1).to(bf)
    xn = torch.zeros(M, 224, device=dev, dtype=bf); xn[:, :N] = (torch.randn(M, N, device=dev) * 0.1).to(bf)
    hout = (torch.randn(M, H, device=dev) * 0.1).to(bf)
    # reference: chunked f32 matmuls accumulated in f64 (time path: h_{t-1} = row - K, masked at t = 0)
    step = (torch.arange(M, device=dev) // K) % T
    hs = torch.zeros_like(hout); hs[K:] = hout[:-K]; hs[step == 0] = 0
    r1 = torch.zeros(4 * H, N, device=dev, dtype=torch.float64); r2 = torch.zeros(4 * H, H, device=dev, dtype=torch.float64)
    for c in range(0, M, 65536):
        a = dg[c:c + 65536].float()
        r1 += (a.t() @ xn[c:c + 65536, :N].float()).double(); r2 += (a.t() @ hs[c:c + 65536].float()).double()
    u = torch.arange(4 * H, device=dev); dst = (u % 4) * H + u // 4            # (unit, gate) rows -> (gate, unit)
    R1 = torch.zeros_like(r1); R2 = torch.zeros_like(r2); R1[dst] = r1; R2[dst] = r2
    scale = max(R1.abs().max().item(), R2.abs().max().item())
    for target in (84, 105, 120, 256):
        for it in range(40):
            g1 = torch.zeros(4 * H, N, device=dev); gb = torch.zeros(4 * H, device=dev); g2 = torch.zeros(4 * H, H, device=dev)
            ops.gemm_tn_dual(dg, xn, g1, gb, hout, g2, 4 * H, N, H, -K, K, T, 0, perm_h=H, target_wgs=target)
            e = max((g1.double() - R1).abs().max().item(), (g2.double() - R2).abs().max().item()) / scale
            worst = max(worst, e)
            assert e < 2e-3, ("outlier", B, target, it, e)
    print("B=%d: ok, worst relative deviation so far %.2e" % (B, worst), flush=True)
