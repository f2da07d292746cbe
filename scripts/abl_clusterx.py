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
    fl = [] if name == "base" else [("-DX" + x) if x.startswith("STAGGER") else ("-DXABL_" + x) for x in name.split("+")]
    so = "/tmp/ablx_%s.so" % name.replace("+", "_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm_clusterx.hip"), os.path.join(CS, "lstm_cluster.hip"), os.path.join(CS, "api.hip"), "-o", so])
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
hx = torch.zeros(plan[4], device=dev, dtype=torch.bfloat16)
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
