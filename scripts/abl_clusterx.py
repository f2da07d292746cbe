"""Ablation timing of the fused cluster forward (csrc/lstm_clusterx.hip; diagnostic builds: results of ablated variants are meaningless).
python scripts/abl_clusterx.py [variant ...]   variant = XABL switches joined by '+', e.g. NO_PROJ+NO_DMA"""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "NO_PROJ", "NO_REC", "NO_CELL", "NO_GATHER", "NO_XSTORE", "NO_GATHER+NO_XSTORE", "NO_DMA", "NO_HSTORE", "NO_HOUT",
                         "NO_DMA+NO_HSTORE+NO_HOUT", "NO_GATHER+NO_XSTORE+NO_DMA+NO_HSTORE+NO_HOUT", "NO_PROJ+NO_REC+NO_CELL"]
libs = {}
for name in names:
    src = os.path.join(CS, "lstm_clusterx.hip")
    if name.startswith("FILE="):                                   # another source file of the same kernel (A/B against a kept copy), no switches
        src, fl = name[5:], []
    else:
        fl = [] if name == "base" else [("-D" + x[2:]) if x.startswith("D:") else ("-DXABL_" + x) for x in name.split("+")]
    so = "/tmp/ablx_%s.so" % name.replace("+", "_").replace("/", "_").replace("=", "_").replace(":", "_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           "-I" + CS, src, os.path.join(CS, "lstm_cluster.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H, Hp, Np = 2 * N, 416, 224
M = B * T * K
dev = "cuda"
xn = torch.randn(M, Np, device=dev).to(torch.bfloat16)
whhq = (torch.randn(2 * 98 * 13 * 512, device=dev) * 0.05).to(torch.bfloat16)
wihq = (torch.randn(2 * 98 * 7 * 512, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(8 * H, device=dev)
gates = torch.empty(M, 8 * H, device=dev, dtype=torch.bfloat16)
hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
c = torch.empty(M, 2 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
plan = (ctypes.c_int64 * 6)()
assert libs[names[0]].urse_lstm_cluster_plan(H, Hp, B * K, 0, plan) == 0
hx = torch.zeros(plan[4] * 27 // 26 + 64, device=dev, dtype=torch.bfloat16)     # rows at the LDS pitch (urse_lstm_clusterx_hx_elems)
cnt = torch.zeros(plan[5], device=dev, dtype=torch.int32)
err = torch.zeros(1, device=dev, dtype=torch.int32)
def fwd(lib):
    return lib.urse_lstm_clusterx_fwd(P(xn.data_ptr()), L(Np), P(wihq.data_ptr()), P(bias.data_ptr()), P(whhq.data_ptr()), P(gates.data_ptr()), L(8 * H),
                                      P(hout.data_ptr()), L(800), P(c.data_ptr()), P(hx.data_ptr()), P(cnt.data_ptr()), P(err.data_ptr()), N, Np, H, Hp,
                                      B * K, T, L(K), L(T * K), L(K), 1, 0, 1, 1, None, P(st))
for name, lib in libs.items():
    assert fwd(lib) == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fwd(lib); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-60s %.3f ms  (%.2f us per step)  err %d" % (name, min(ts), min(ts) * 1e3 / T, int(err.item())), flush=True)
    err.zero_()

# in-kernel cycle stamps of one workgroup (variants built with D:XSTAMP=<workgroup>): median cycles between the stamps over the steps
import numpy as np
for name, lib in libs.items():
    if "XSTAMP" not in name:
        continue
    fwd(lib); torch.cuda.synchronize()
    buf = np.zeros(512 * 16, dtype=np.uint64)
    assert lib.urse_diag_clusterx_stamps(buf.ctypes.data_as(P)) == 0
    sa = buf.reshape(512, 16)[5:T - 2].astype(np.int64)
    nx = buf.reshape(512, 16)[6:T - 1].astype(np.int64)
    lab = ["top", "issued+projected", "gathered", "at b1", "after b1", "at b2", "after b2", "published"]
    print(name, "working wave 0, median shader-clock cycles:")
    for i in range(1, 8):
        print("   %-18s -> %-18s %8.0f" % (lab[i - 1], lab[i], np.median(sa[:, i] - sa[:, i - 1])))
    print("   %-18s -> %-18s %8.0f" % ("published", "next top", np.median(nx[:, 0] - sa[:, 7])))
    print("   step                                   %8.0f" % np.median(nx[:, 0] - sa[:, 0]))
    if np.median(sa[:, 12]) > 0:      # phase 2 by parts (XPIPE): head = row tile 0's MFMAs, three interleaved phases, tail = row tile 3's cell update
        print("   phase 2: head %.0f | tile 1 MFMAs + tile 0 cells %.0f | tile 2 + 1 %.0f | tile 3 + 2 %.0f | tail (tile 3 cells) %.0f" % (
            np.median(sa[:, 12] - sa[:, 4]), np.median(sa[:, 13] - sa[:, 12]), np.median(sa[:, 14] - sa[:, 13]), np.median(sa[:, 15] - sa[:, 14]), np.median(sa[:, 5] - sa[:, 15])))
    print("   helper: arrives at b1 %.0f cycles after worker 0 | b1 -> DMAs issued %.0f | -> landed %.0f | arrives at b2 %.0f after worker 0 | b2 -> next b1 arrival (pieces read + stored) %.0f" % (
        np.median(sa[:, 8] - sa[:, 3]), np.median(sa[:, 10] - sa[:, 9]), np.median(sa[:, 11] - sa[:, 10]), np.median(sa[:, 11] - sa[:, 5]), np.median(nx[:, 8] - sa[:, 11])))
