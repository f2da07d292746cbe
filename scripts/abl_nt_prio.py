"""A/B of s_setprio around the NT ring kernel's MFMA cluster (two builds), on the step's NT shapes."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
libs = {}
for name, fl in {"prio": ["-DURSE_NT_SETPRIO=1"], "plain": ["-DURSE_NT_SETPRIO=0"]}.items():
    so = "/tmp/ablnp_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "gemm.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
dev, bf = "cuda", torch.bfloat16
M = 32 * 401 * 34
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
for name, N, K, odt in (("ih fwd", 3136, 224, 1), ("dgrad ih", 224, 3136, 0), ("fc fwd", 196, 800, 0), ("dgrad fc", 800, 224, 1)):
    a = (torch.randn(M, K, device=dev) * 0.1).to(bf)
    w = (torch.randn(N, K, device=dev) * 0.1).to(bf)
    c = torch.empty(M, N, device=dev, dtype=bf if odt == 1 else torch.float32)
    res = []
    for ln, lib in libs.items():
        run = lambda: lib.urse_gemm_nt(P(a.data_ptr()), L(K), P(w.data_ptr()), L(K), P(c.data_ptr()), L(N), P(0), P(0), L(0),
                                       L(M), L(N), L(K), 1, odt, 0, P(st))
        assert run() == 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): run()
        torch.cuda.synchronize()
        res.append("%s %.3f" % (ln, (time.perf_counter() - t0) / 5 * 1e3))
    print("%-9s N=%d K=%d: %s ms" % (name, N, K, " | ".join(res)), flush=True)
