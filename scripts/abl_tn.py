"""A/B of the TN ring depth (diagnostic build)."""
import ctypes, os, subprocess, sys, time
import torch
TN_TARGET = [0]      # urse_gemm_tn's per-call target_workgroups (0 = one per CU)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"nst4": [], "nst5": ["-DURSE_TN_NST=5"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/abltn_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
M, N, H = 32 * 401 * 34, 196, 392
dev, bf = "cuda", torch.bfloat16
dg = (torch.randn(M, 8 * H, device=dev) * 0.1).to(bf)
xn = (torch.randn(M, 224, device=dev) * 0.1).to(bf)
gw = torch.zeros(8 * H, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def run(lib):
    return lib.urse_gemm_tn(P(dg.data_ptr()), L(8 * H), P(xn.data_ptr()), L(224), P(gw.data_ptr()), L(N), P(0), L(M), L(8 * H), L(N),
                            L(0), L(1), L(0), L(0), L(H), 1, TN_TARGET[0], P(st))
res = []
for name, lib in libs.items():
    assert run(lib) == 0, name
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(lib)
    torch.cuda.synchronize()
    res.append("%s %.3f" % (name, (time.perf_counter() - t0) / 5 * 1e3))
print("tn wih:", " | ".join(res), "ms", flush=True)
