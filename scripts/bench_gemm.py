"""Times the GEMM shapes of one dual-path half layer at the C2 config (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
M, N, H = 32 * 401 * 34, 196, 392
dev, bf = "cuda", torch.bfloat16
r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(bf)
xn, wih, gx = r(M, 224), r(8 * H, 224), torch.empty(M, 8 * H, device=dev, dtype=bf)
bias = torch.randn(8 * H, device=dev)
hout, wfc, skip = r(M, 800), r(N, 800), torch.randn(M, N, device=dev)
out = torch.empty(M, N, device=dev)
dg, wihT = r(M, 8 * H), r(N, 8 * H)
doT, wfcT, dh = r(M, 224), r(2 * H, 224), torch.empty(M, 800, device=dev, dtype=bf)
gwih, gwhh, gwfc = torch.zeros(8 * H, N, device=dev), torch.zeros(4 * H, H, device=dev), torch.zeros(N, 2 * H, device=dev)
cs = torch.zeros(8 * H, device=dev)
def t(name, fn, flops, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-28s %7.3f ms  %7.1f TF/s" % (name, dt * 1e3, flops / dt / 1e12), flush=True)
t("nt ih fwd  (N3136 K224)", lambda: ops.gemm_nt(xn, wih, bias, out=gx), 2.0 * M * 8 * H * 196)
t("nt fc fwd  (N196 K800 res)", lambda: ops.gemm_nt(hout, wfc, bias[:N], resid=skip, out=out), 2.0 * M * N * 784)
t("nt dgrad ih (N196 K3136)", lambda: ops.gemm_nt(dg, wihT, out=out), 2.0 * M * N * 8 * H)
t("nt dgrad fc (N784 K224)", lambda: ops.gemm_nt(doT, wfcT, out=dh, N=2 * H), 2.0 * M * 2 * H * 196)
t("tn wih (3136x196)", lambda: ops.gemm_tn(dg, xn, gwih, colsum=cs, Mo=8 * H, No=N, perm_h=H), 2.0 * M * 8 * H * 196)
t("tn whh (1568x392 shifted)", lambda: ops.gemm_tn(dg[:, :4 * H], hout[:, :H], gwhh, Mo=4 * H, No=H, shift=-34, inner=34, period=401, invalid_step=0, perm_h=H), 2.0 * M * 4 * H * H)
t("tn wfc (196x784)", lambda: ops.gemm_tn(doT, hout, gwfc, colsum=cs[:N], Mo=N, No=2 * H), 2.0 * M * N * 2 * H)
# library yardstick (hipBLASLt / rocBLAS through torch) for the same contractions; not used by the product path
gxo = torch.empty(M, 8 * H, device=dev, dtype=bf)
t("torch ih fwd", lambda: torch.mm(xn, wih.t(), out=gxo), 2.0 * M * 8 * H * 196)
o2 = torch.empty(M, N, device=dev, dtype=bf)
t("torch fc fwd", lambda: torch.mm(hout, wfc.t(), out=o2), 2.0 * M * N * 784)
t("torch dgrad ih", lambda: torch.mm(dg, wihT.t(), out=o2), 2.0 * M * N * 8 * H)
o3 = torch.empty(M, 2 * H, device=dev, dtype=bf)
t("torch dgrad fc", lambda: torch.mm(doT, wfcT.t(), out=o3), 2.0 * M * 2 * H * 196)
o4 = torch.empty(8 * H, 224, device=dev, dtype=bf)
t("torch tn wih", lambda: torch.mm(dg.t(), xn, out=o4), 2.0 * M * 8 * H * 196)
o5 = torch.empty(4 * H, H, device=dev, dtype=bf)
hh = hout[:, :H].contiguous()
t("torch tn whh", lambda: torch.mm(dg[:, :4 * H].t(), hh, out=o5), 2.0 * M * 4 * H * H)
gw1, gw2 = torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, H, device=dev)
t("tn dual (wih_d + whh_d)", lambda: ops.gemm_tn_dual(dg[:, :4 * H], xn, gw1, cs[:4 * H], hout[:, :H], gw2, 4 * H, N, H, -34, 34, 401, 0, perm_h=H),
  2.0 * M * 4 * H * (196 + H))
# dgrad + GroupNorm backward: two-pass (GEMM, reduce + apply) against the reduce on the GEMM's epilogue (urse_gemm_nt_gnbwd)
Bn, rows = 32, 401 * 34
xs = torch.randn(Bn, 401, 34, N, device=dev)
gam, bet = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
_, stats = ops.groupnorm_fwd(xs, gam, bet, Bn, 401, 1, 34 * N, N, 224, 0, bf)
dgm, dbt, dres = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.randn(Bn, 401, 34, N, device=dev)
def two_pass():
    dy = ops.gemm_nt(dg, wihT, out_dtype=torch.float32, N=N)
    return ops.groupnorm_bwd(xs, dy, stats, gam, dres, dgm, dbt, Bn, 401, 1, 34 * N, N, 0, pack_ld=224)
def fused():
    dy, sm = ops.gemm_nt_gnbwd(dg, wihT, N, xs, stats, gam, dgm, dbt, rows)
    return ops.groupnorm_bwd(xs, dy, stats, gam, dres, dgm, dbt, Bn, 401, 1, 34 * N, N, 0, pack_ld=224, sums=sm)
t("dgrad + GN bwd, two-pass", two_pass, 2.0 * M * N * 8 * H)
t("dgrad + GN bwd, fused reduce", fused, 2.0 * M * N * 8 * H)
t("  fused GEMM alone", lambda: ops.gemm_nt_gnbwd(dg, wihT, N, xs, stats, gam, dgm, dbt, rows), 2.0 * M * N * 8 * H)
