"""A/B of the TN ring kernel's DMA issue path (URSE_TN_LEAN_ISSUE 0 | 1): the dual-operand weight gradient of one LSTM direction
at C2 (dW_ih + bias + dW_hh from one pass over the [M, 4H] dgates), time-path and band-path row maps; checks that both builds
give the same gradients."""
import ctypes, os, subprocess, sys, time
import torch
TN_TARGET = [0]      # urse_gemm_tn's per-call target_workgroups (0 = one per CU)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"generic": ["-DURSE_TN_LEAN_ISSUE=0"], "lean": ["-DURSE_TN_LEAN_ISSUE=1"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/abltni_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N, H = 32, 401, 34, 196, 392
M = B * T * K
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
dg = (torch.randn(M, 8 * H, device=dev) * 0.1).to(bf)
xn = (torch.randn(M, 224, device=dev) * 0.1).to(bf)
xn[:, N:] = 0
hout = (torch.randn(M, 800, device=dev) * 0.1).to(bf)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def run(lib, dr, path, outs):
    gwih, gb, gwhh = outs
    stride, seq = (K, T) if path == "t" else (1, K)
    sh, inv = (-stride, 0) if dr == 0 else (stride, seq - 1)
    a = dg[:, dr * 4 * H:(dr + 1) * 4 * H]
    h = hout[:, dr * H:(dr + 1) * H]
    return lib.urse_gemm_tn_dual(P(a.data_ptr()), L(8 * H), P(xn.data_ptr()), L(224), P(gwih.data_ptr()), L(N), P(gb.data_ptr()),
                                 P(h.data_ptr()), L(800), P(gwhh.data_ptr()), L(H), L(M), L(4 * H), L(N), L(H), L(sh), L(stride),
                                 L(seq), L(inv), L(H), 1, TN_TARGET[0], P(st))
ref = {}
for path in ("t", "f"):
    res = []
    for name, lib in libs.items():
        outs = (torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(4 * H, H, device=dev))
        for dr in (0, 1):
            assert run(lib, dr, path, outs) == 0, (name, lib.urse_last_error)
        torch.cuda.synchronize()
        key = (path,)
        if key in ref:
            d = max(float((a - b).abs().max() / (b.abs().max() + 1e-9)) for a, b in zip(outs, ref[key]))
            res.append("max rel diff vs generic %.2e" % d)
        else:
            ref[key] = [o.clone() for o in outs]
        t0 = time.perf_counter()
        for _ in range(5):
            run(lib, 0, path, outs); run(lib, 1, path, outs)
        torch.cuda.synchronize()
        res.append("%s %.3f ms per dual call" % (name, (time.perf_counter() - t0) / 10 * 1e3))
    print("path %s:" % path, " | ".join(res), flush=True)
