"""Ablation timing of the cluster LSTM forward (diagnostic builds; results of ablated variants are meaningless)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "NO_GATHER", "NO_MFMA", "NO_CELL", "NO_DEFERRED", "NO_GX", "NO_XSTORE", "NO_GATHER+NO_XSTORE", "NO_DEFERRED+NO_GX",
                         "NO_GATHER+NO_XSTORE+NO_DEFERRED+NO_GX", "NO_GATHER+NO_XSTORE+NO_DEFERRED+NO_GX+NO_MFMA+NO_CELL"]
variants = {n: ([] if n == "base" else ["-DCABL_" + x for x in n.split("+")]) for n in names}
libs = {}
for name, fl in variants.items():
    so = "/tmp/ablc_%s.so" % name.replace("+", "_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm_cluster.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H, Hp = 2 * N, 416
M = B * T * K
dev = "cuda"
gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
whhq = (torch.randn(2 * 98 * 13 * 512, device=dev) * 0.05).to(torch.bfloat16)
hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
c = torch.empty(M, 2 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
plan = (ctypes.c_int64 * 6)()
assert libs["base"].urse_lstm_cluster_plan(H, Hp, B * K, 0, plan) == 0
print("plan", list(plan))
hx = torch.zeros(plan[4], device=dev, dtype=torch.bfloat16)
cnt = torch.zeros(plan[5], device=dev, dtype=torch.int32)
err = torch.zeros(1, device=dev, dtype=torch.int32)
def fwd(lib):
    return lib.urse_lstm_cluster_fwd(P(gx.data_ptr()), ctypes.c_int64(8 * H), P(whhq.data_ptr()), P(hout.data_ptr()), ctypes.c_int64(800),
        P(c.data_ptr()), P(hx.data_ptr()), P(cnt.data_ptr()), P(err.data_ptr()), H, Hp, B * K, T, ctypes.c_int64(K), ctypes.c_int64(T * K),
        ctypes.c_int64(K), 1, 0, 1, 1, None, P(st))
for name, lib in libs.items():
    assert fwd(lib) == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fwd(lib); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-60s %.3f ms  (%.2f us per step)  err %d" % (name, min(ts), min(ts) * 1e3 / T, int(err.item())), flush=True)
