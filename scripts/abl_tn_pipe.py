"""A/B of the TN ring kernel's k loop: URSE_TN_PIPE variants (0 flat, 2 two stages per barrier, 4 = 2 + DMA issue between the MFMA groups), on
one LSTM direction's dual wgrad at C2; also checks the two builds agree bit for bit on a fixed input."""
import ctypes, os, subprocess, sys, time
import torch
TN_TARGET = [0]      # urse_gemm_tn's per-call target_workgroups (0 = one per CU)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
libs = {}
for name, fl in {"flat": ["-DURSE_TN_PIPE=0"], "pair": ["-DURSE_TN_PIPE=2"], "pipe": ["-DURSE_TN_PIPE=4"]}.items():
    so = "/tmp/abltp_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
H, N, B, T, K = 392, 196, 32, 401, 34
M = B * T * K
dev, bf = "cuda", torch.bfloat16
dg = (torch.randn(M, 8 * H, device=dev) * 0.1).to(bf)
xn = torch.zeros(M, 224, device=dev, dtype=bf); xn[:, :N] = (torch.randn(M, N, device=dev) * 0.1).to(bf)
hout = (torch.randn(M, 2 * H, device=dev) * 0.1).to(bf)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def run(lib, gwih, gb, gwhh):
    A = dg[:, :4 * H]
    B2 = hout[:, :H]
    rc = lib.urse_gemm_tn_dual(P(A.data_ptr()), L(A.stride(0)), P(xn.data_ptr()), L(224), P(gwih.data_ptr()), L(N), P(gb.data_ptr()),
                               P(B2.data_ptr()), L(B2.stride(0)), P(gwhh.data_ptr()), L(H), L(M), L(4 * H), L(N), L(H),
                               L(-K), L(K), L(T), L(0), L(H), 1, TN_TARGET[0], P(st))
    assert rc == 0
outs = {}
for target in (105, 252):
    res = []
    for name, lib in libs.items():
        TN_TARGET[0] = target
        g = [torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(4 * H, H, device=dev)]
        run(lib, *g); torch.cuda.synchronize()
        outs[name] = [x.clone() for x in g]
        t0 = time.perf_counter()
        for _ in range(5): run(lib, *g)
        torch.cuda.synchronize()
        res.append("%s %.3f" % (name, (time.perf_counter() - t0) / 5 * 1e3))
    ref = dg[:, :4 * H].float().t() @ xn[:, :N].float()
    err = (outs["pipe"][0] - ref).abs().max().item() / ref.abs().max().item()
    d = max(max((a - b).abs().max().item() for a, b in zip(outs[x], outs["flat"])) for x in ("pipe", "pair"))
    print("target %d: %s ms   rel err vs f32 matmul (perm ignored) %.2e   max |pipe/pair - flat| %.3g" % (target, " | ".join(res), err, d), flush=True)
