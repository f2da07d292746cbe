"""Ablation timing of the N-split BPTT (diagnostic builds of csrc/experiments/lstm_nsplit3.hip: the round-5 three-member N-split, not part of the shipped library; wrong results, timing only)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "D:N3_PUB=0", "NO_STORE", "NO_POLL", "NO_COPY", "NO_LOAD", "NO_MM"]
libs = {}
for name in names:
    fl = [] if name == "base" else [("-D" + x[2:]) if x.startswith("D:") else ("-DN3ABL_" + x) for x in name.split("+")]
    so = "/tmp/ablns3_%s.so" % name.replace("+", "_").replace("=", "").replace(":", "")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "experiments", "lstm_nsplit3.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H = 2 * N
M = B * T * K
dev = "cuda"
g0 = torch.rand(M, 8 * H, device=dev).to(torch.bfloat16)
c = torch.randn(M, 2 * H, device=dev)
whhT = (torch.randn(2 * 400 * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
dh = (0.1 * torch.randn(M, 800, device=dev)).to(torch.bfloat16)
flags = torch.zeros(4096, device=dev, dtype=torch.int32)
err = torch.zeros(1, device=dev, dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def bwd(lib, g):
    rc = lib.urse_lstm_nsplit3_bwd(P(dh.data_ptr()), L(800), P(g.data_ptr()), L(8 * H), P(c.data_ptr()), P(whhT.data_ptr()), P(flags.data_ptr()),
                                  P(err.data_ptr()), H, B * K, T, L(K), L(T * K), L(K), 0, P(st))
    assert rc == 0, rc
g = g0.clone()
for name, lib in libs.items():
    bwd(lib, g); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        g.copy_(g0); torch.cuda.synchronize()
        t0 = time.perf_counter(); bwd(lib, g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %.3f ms  (%.2f us per step)  err %d" % (name, min(ts), min(ts) * 1e3 / T, int(err.item())), flush=True)
