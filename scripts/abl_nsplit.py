"""Ablation timing of the N-split BPTT (diagnostic builds of csrc/lstm_nsplit.hip; wrong results, timing only)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "PUB=0", "PUB=2", "NO_STORE", "NO_POLL", "NO_LOAD"]
libs = {}
for name in names:
    fl = [] if name == "base" else [("-DNS_" + x) if x.startswith("PUB") else ("-D" + x[2:]) if x.startswith("D:") else ("-DNSABL_" + x) for x in name.split("+") if x != "WIDE"]
    so = "/tmp/ablns_%s.so" % name.replace("+", "_").replace("=", "").replace(":", "")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm_nsplit.hip"), os.path.join(CS, "api.hip"), "-o", so])
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
    rc = lib.urse_lstm_nsplit_bwd(P(dh.data_ptr()), L(800), P(g.data_ptr()), L(8 * H), P(c.data_ptr()), P(whhT.data_ptr()), P(flags.data_ptr()),
                                  P(err.data_ptr()), H, B * K, T, L(K), L(T * K), L(K), 0, P(st))
    assert rc == 0, rc
g = g0.clone()
for name, lib in libs.items():
    os.environ["URSE_NSPLIT_WIDE"] = "1" if "WIDE" in name.split("+") else "0"      # the seven-wave / two-tile kernel of the same entry point
    bwd(lib, g); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        g.copy_(g0); torch.cuda.synchronize()
        t0 = time.perf_counter(); bwd(lib, g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %.3f ms  (%.2f us per step)  err %d" % (name, min(ts), min(ts) * 1e3 / T, int(err.item())), flush=True)

# in-kernel cycle stamps (variants built with D:NSSTAMP=<workgroup>): median cycles between the stamps of waves 0 and 6
import numpy as np
for name, lib in libs.items():
    if "NSSTAMP" not in name:
        continue
    os.environ["URSE_NSPLIT_WIDE"] = "0"
    g.copy_(g0); bwd(lib, g); torch.cuda.synchronize()
    buf = np.zeros(512 * 16, dtype=np.uint64)
    assert lib.urse_diag_nsplit_stamps(buf.ctypes.data_as(P)) == 0
    sa = buf.reshape(512, 2, 8)[5:T - 3].astype(np.int64); nx = buf.reshape(512, 2, 8)[6:T - 2].astype(np.int64)
    lab = ["top", "cell phase done", "barrier 1 passed", "own half stored (issued)", "own product done", "partner flag + sync", "copy + barrier", "other product done"]
    for wv, wn in ((0, "wave 0"), (1, "wave 6")):
        print(name, wn, "median shader-clock cycles:")
        for i in range(1, 8):
            print("   %-26s -> %-26s %8.0f" % (lab[i - 1], lab[i], np.median(sa[:, wv, i] - sa[:, wv, i - 1])))
        print("   step %8.0f" % np.median(nx[:, wv, 0] - sa[:, wv, 0]))
