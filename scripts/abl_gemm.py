"""Ablation timing of the LDS-DMA NT GEMM (diagnostic builds; ablated variants compute garbage)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"base": [], "direct": ["-DURSE_NT_DIRECT_EPILOGUE"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/ablg_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
M, N, H = 32 * 401 * 34, 196, 392
dev, bf = "cuda", torch.bfloat16
xn = (torch.randn(M, 224, device=dev) * 0.1).to(bf)
wih = (torch.randn(8 * H, 224, device=dev) * 0.1).to(bf)
gx = torch.empty(M, 8 * H, device=dev, dtype=bf)
bias = torch.randn(8 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def run(lib):
    return lib.urse_gemm_nt(P(xn.data_ptr()), L(224), P(wih.data_ptr()), L(224), P(gx.data_ptr()), L(8 * H), P(bias.data_ptr()), P(0), L(0),
                            L(M), L(8 * H), L(224), 1, 1, 0, P(st))
res = []
for name, lib in libs.items():
    assert run(lib) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): run(lib)
    torch.cuda.synchronize()
    res.append("%s %.3f" % (name, (time.perf_counter() - t0) / 3 * 1e3))
print("nt ih fwd:", " | ".join(res), "ms", flush=True)
