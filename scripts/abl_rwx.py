"""Ablation timing of the fused row-wave forward (diagnostic builds of csrc/lstm_rwx.hip; wrong results, timing only)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
names = sys.argv[1:] or ["base", "NO_DMA_WAIT", "NO_DMA", "NO_MFMA", "NO_CELL", "CHEAP_CELL", "NO_STORE", "NO_CLOAD", "NO_STORE+NO_CLOAD",
                         "NO_STORE+NO_CLOAD+NO_DMA_WAIT", "NO_MFMA+CHEAP_CELL", "NO_MFMA+CHEAP_CELL+NO_DMA", "NO_STORE+NO_CLOAD+NO_CELL", "NO_STORE+NO_CLOAD+NO_CELL+NO_DMA_WAIT"]
libs = {}
for name in names:
    fl = [] if name == "base" else [("-D" + x[2:]) if x.startswith("D:") else ("-DRXABL_" + x) for x in name.split("+")]      # D:MACRO=v passes a plain define
    so = "/tmp/ablrwx_%s.so" % name.replace("+", "_").replace(":", "_").replace("=", "_")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm_rwx.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H, Hp, Np = 2 * N, 416, 224
M = B * T * K
dev = "cuda"
xn = torch.randn(M, Np, device=dev).to(torch.bfloat16)
wx = (torch.randn(2 * 25 * 20 * 4 * 512, device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn(8 * H, device=dev)
gates = torch.empty(M, 8 * H, device=dev, dtype=torch.bfloat16)
hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
c = torch.zeros(M, 2 * H, device=dev)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
def fwd(lib):
    rc = lib.urse_lstm_rwx_fwd(P(xn.data_ptr()), L(Np), P(wx.data_ptr()), P(bias.data_ptr()), P(gates.data_ptr()), L(8 * H), P(hout.data_ptr()), L(800),
                               P(c.data_ptr()), N, Np, H, Hp, B * T, K, L(1), L(K), L(1), 1, 0, 1, None, P(st))
    assert rc == 0, rc
for name, lib in libs.items():
    fwd(lib); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fwd(lib); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-44s %.3f ms  (%.1f us per step)" % (name, min(ts), min(ts) * 1e3 / K), flush=True)
