"""Times the per-band (grouped) NT GEMMs of the mask decoder at C2: 34 bands x 12,832 rows (diagnostic)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops, bsrnn
from urgent2026_challenge_track1_amd._lib import call, stream_ptr
G, M = 34, 32 * 401
dev, bf = "cuda", torch.bfloat16
def mk(*s): return (torch.randn(*s, device=dev) * 0.1).to(bf)
def rows(A, B, C, bias, resid, M, N, K):
    out = []
    for g in range(G):
        out.append([A[g].data_ptr(), B[g].data_ptr(), C[g].data_ptr(), bias[g].data_ptr() if bias is not None else 0,
                    resid[g].data_ptr() if resid is not None else 0, A.stride(1), B.stride(1), C.stride(1), M, N, K, resid.stride(1) if resid is not None else 0])
    return out
def t(name, rws, in_dt, out_dt, act, flops, nbytes, n=5):
    f = lambda: bsrnn.nt_grouped(rws, dev, in_dt, out_dt, act)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%-44s %7.3f ms  %6.1f TF/s  %5.2f TB/s" % (name, dt * 1e3, flops / dt / 1e12, nbytes / dt / 1e12), flush=True)
BF, F32 = 1, 0
import ctypes
lib = __import__("urgent2026_challenge_track1_amd._lib", fromlist=["x"]).load()
X = mk(G, M, 224); W1 = mk(G, 784, 224); H1 = torch.empty(G, M, 800, device=dev, dtype=bf); b1 = torch.randn(G, 784, device=dev)
t("mask fc1 + tanh (K224 N784, bf16 out)", rows(X, W1, H1, b1, None, M, 784, 224), ops._dt(X), ops._dt(H1), 1, 2.0 * G * M * 784 * 196, G * M * (224 + 800) * 2)
t("mask fc1 no act", rows(X, W1, H1, b1, None, M, 784, 224), ops._dt(X), ops._dt(H1), 0, 2.0 * G * M * 784 * 196, G * M * (224 + 800) * 2)
W2 = mk(G, 224, 800); O2 = torch.empty(G, M, 224, device=dev); b2 = torch.randn(G, 224, device=dev)
t("mask fc2 (K800 N<=224, f32 out)", rows(H1, W2, O2, b2, None, M, 224, 800), ops._dt(H1), ops._dt(O2), 0, 2.0 * G * M * 224 * 784, G * M * (800 * 2 + 224 * 4))
DY = mk(G, M, 800); W1T = mk(G, 224, 800); DX = torch.empty(G, M, 224, device=dev, dtype=bf)
t("fc1 dgrad (K800 N224, bf16 out)", rows(DY, W1T, DX, None, None, M, 224, 800), ops._dt(DY), ops._dt(DX), 0, 2.0 * G * M * 224 * 784, G * M * (800 + 224) * 2)
DO = mk(G, M, 224); W2T = mk(G, 784, 224); DH = torch.empty(G, M, 800, device=dev, dtype=bf)
t("fc2 dgrad + tanh-bwd (K224 N784, resid=h)", rows(DO, W2T, DH, None, H1, M, 784, 224), ops._dt(DO), ops._dt(DH), 2, 2.0 * G * M * 784 * 196, G * M * (224 + 800 + 800) * 2)
