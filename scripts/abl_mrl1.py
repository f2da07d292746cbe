"""Ablation timing of the MR-L1 spectral kernels (diagnostic builds of loss.hip): what the 4 x 160 us per step are made of."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"base": [], "plain_stores": ["-DMABL_NO_ATOMIC"], "no_scatter": ["-DMABL_NO_SCATTER"], "no_fft": ["-DMABL_NO_FFT"],
            "no_fft_no_scatter": ["-DMABL_NO_FFT", "-DMABL_NO_SCATTER"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/ablm_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-Wno-unused-value", *fl,
                           os.path.join(CS, "loss.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, L = 32, 192000
t = torch.randn(B, L, device="cuda"); e = t + 0.3 * torch.randn(B, L, device="cuda")
loss = torch.empty(B, device="cuda"); G = torch.empty(B, L, device="cuda")
sums = torch.empty(B * 5, device="cuda", dtype=torch.float64); acc = torch.empty(B * 2, device="cuda", dtype=torch.float64)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
for wins in ((256,), (512,), (768,), (1024,), (256, 512, 768, 1024)):
    w = (ctypes.c_int32 * len(wins))(*wins)
    res = []
    for name, lib in list(libs.items()) + [("lds_pass_kernel", libs["base"])]:
        os.environ.pop("URSE_MRL1_NO_REG_FFT", None)
        if name == "lds_pass_kernel": os.environ["URSE_MRL1_NO_REG_FFT"] = "1"
        run = lambda: lib.urse_mrl1_loss_fwd(P(t.data_ptr()), P(e.data_ptr()), P(loss.data_ptr()), P(G.data_ptr()), P(sums.data_ptr()),
                                             P(acc.data_ptr()), B, L, w, len(wins), ctypes.c_float(1e-6), ctypes.c_float(0.5), P(st))
        assert run() == 0, name
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): run()
        b.record(); torch.cuda.synchronize()
        res.append("%s %.0f" % (name, a.elapsed_time(b) / 20 * 1e3))
    print("windows", wins, "| fwd incl. pair sums + time-domain term:", " | ".join(res), "us", flush=True)
