"""Ablation timing of the row-wave LSTM forward (diagnostic builds of csrc/lstm_rw.hip; wrong results, timing only)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "NO_MFMA+CHEAP_CELL", "NO_MFMA+CHEAP_CELL+NO_DMA", "CHEAP_CELL", "NO_STORE+NO_HOUT", "NO_LOAD+NO_HOUT", "NO_LOAD+NO_STORE",
                         "NO_GLOAD", "NO_CLOAD", "NO_CSTORE", "NO_GSTORE", "NO_LOAD+NO_STORE+NO_HOUT", "NO_MFMA+CHEAP_CELL+NO_STORE+NO_HOUT",
                         "NO_MFMA+CHEAP_CELL+NO_LOAD+NO_HOUT", "NO_MFMA+CHEAP_CELL+NO_LOAD+NO_STORE+NO_HOUT"]
libs = {}
for name in names:
    fl = [] if name == "base" else ["-DRWABL_" + x for x in name.split("+")]
    so = "/tmp/ablrw_%s.so" % name.replace("+", "_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm_rw.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H, Hp = 2 * N, 416
M = B * T * K
dev = "cuda"
gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
whhb = (torch.randn(2 * 25 * 13 * 4 * 512, device=dev) * 0.05).to(torch.bfloat16)
hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
c = torch.zeros(M, 2 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def fwd(lib):
    rc = lib.urse_lstm_rw_fwd(P(gx.data_ptr()), L(8 * H), P(whhb.data_ptr()), P(hout.data_ptr()), L(800), P(c.data_ptr()), H, Hp,
                              B * T, K, L(1), L(K), L(1), 1, 0, P(st))
    assert rc == 0, rc
for name, lib in libs.items():
    fwd(lib); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fwd(lib); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %.3f ms  (%.1f us per step)" % (name, min(ts), min(ts) * 1e3 / K), flush=True)
