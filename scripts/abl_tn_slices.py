"""Isolated dual TN wgrad (one LSTM direction at C2) against the workgroup target: does the per-CU stream rate depend on
how the R slices fall onto the XCDs (xcd_remap gives XCD x the consecutive ids x*n/8 ..)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
from urgent2026_challenge_track1_amd._lib import call
dev, bf = "cuda", torch.bfloat16
H, N, B, T, K = 392, 196, 32, 401, 34
M = B * T * K
dg = (torch.randn(M, 8 * H, device=dev) * 0.1).to(bf)
xn = torch.zeros(M, 224, device=dev, dtype=bf); xn[:, :N] = (torch.randn(M, N, device=dev) * 0.1).to(bf)
hout = (torch.randn(M, 2 * H, device=dev) * 0.1).to(bf)
gwih = torch.zeros(4 * H, N, device=dev); gb = torch.zeros(4 * H, device=dev); gwhh = torch.zeros(4 * H, H, device=dev)
def run(target=0):
    ops.gemm_tn_dual(dg[:, :4 * H], xn, gwih, gb, hout[:, :H], gwhh, 4 * H, N, H, -K, K, T, 0, perm_h=H, target_wgs=target)
for target in [int(a) for a in sys.argv[1:]] or [105, 126, 147, 168, 210, 252, 336]:
    run(target); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(target)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    slices = target // 21
    wg_mb = M / slices * 480 * 2 / 1e6
    print("target %3d  slices %2d  %.3f ms   per-WG %.1f MB -> %.1f GB/s per WG" % (target, slices, ms, wg_mb, wg_mb / ms), flush=True)
