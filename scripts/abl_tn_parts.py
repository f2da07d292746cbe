"""Ablation timing of the TN ring kernel's flat k loop (diagnostic builds; ablated variants compute garbage): which of
DMA / fragment reads / MFMA / barrier sets the 1 us k-step?"""
import ctypes, os, subprocess, sys, time
import torch
TN_TARGET = [0]      # urse_gemm_tn's per-call target_workgroups (0 = one per CU)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"base": [], "no_mfma": ["-DTABL_NO_MFMA"], "no_read": ["-DTABL_NO_READ"], "no_dma": ["-DTABL_NO_DMA"],
            "zero_dma": ["-DTABL_ZERO_DMA"], "no_barrier": ["-DTABL_NO_BARRIER"],
            "no_read_no_mfma": ["-DTABL_NO_READ", "-DTABL_NO_MFMA"], "no_dma_no_read": ["-DTABL_NO_DMA", "-DTABL_NO_READ"],
            "no_dma_no_mfma": ["-DTABL_NO_DMA", "-DTABL_NO_MFMA"]}
if len(sys.argv) > 1:
    variants = {k: v for k, v in variants.items() if k in sys.argv[1:] or k == "base"}
libs = {}
for name, fl in variants.items():
    so = "/tmp/abltq_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-DURSE_TN_PIPE=0",
                           *fl, os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
H, N, B, T, K = 392, 196, 32, 401, 34
M = B * T * K
dev, bf = "cuda", torch.bfloat16
dg = (torch.randn(M, 8 * H, device=dev) * 0.1).to(bf)
xn = torch.zeros(M, 224, device=dev, dtype=bf); xn[:, :N] = (torch.randn(M, N, device=dev) * 0.1).to(bf)
hout = (torch.randn(M, 2 * H, device=dev) * 0.1).to(bf)
g = [torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, device=dev), torch.zeros(4 * H, H, device=dev)]
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def run(lib):
    A = dg[:, :4 * H]; B2 = hout[:, :H]
    rc = lib.urse_gemm_tn_dual(P(A.data_ptr()), L(A.stride(0)), P(xn.data_ptr()), L(224), P(g[0].data_ptr()), L(N), P(g[1].data_ptr()),
                               P(B2.data_ptr()), L(B2.stride(0)), P(g[2].data_ptr()), L(H), L(M), L(4 * H), L(N), L(H),
                               L(-K), L(K), L(T), L(0), L(H), 1, TN_TARGET[0], P(st))
    assert rc == 0
for name, lib in libs.items():
    TN_TARGET[0] = 105
    run(lib); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run(lib)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("%-18s %.3f ms  (%.0f ns per k-step)" % (name, ms, ms * 1e6 / (M / 5 / 32)), flush=True)
