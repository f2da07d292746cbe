"""Ablation timing of the 960-point STFT kernel (diagnostic builds)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"base": [], "nodft": ["-DSTABL_NO_DFT"], "nostore": ["-DSTABL_NO_STORE"], "neither": ["-DSTABL_NO_DFT", "-DSTABL_NO_STORE"], "nff16": ["-DURSE_STFT960_NFF=16"], "nff4": ["-DURSE_STFT960_NFF=4"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/ablst_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "stft.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, L = 32, 192000
x = torch.randn(B, L, device="cuda")
spec = torch.empty(B, 401, 481, 2, device="cuda")
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
def run(lib):
    return lib.urse_stft_fwd(P(x.data_ptr()), P(0), P(spec.data_ptr()), B, L, 960, 480, 1, P(st))
for Bn in (32,):
    B = Bn
    x = torch.randn(B, L, device="cuda")
    spec = torch.empty(B, 401, 481, 2, device="cuda")
    res = []
    for name, lib in libs.items():
        nf = 960 if name != "generic" else 960
        assert run(lib) == 0, name
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): run(lib)
        b.record(); torch.cuda.synchronize()
        res.append("%s %.1f" % (name, a.elapsed_time(b) / 50 * 1e3))
    print("B", B, "stft960:", " | ".join(res), "us", flush=True)
