"""Ablation timing of the LSTM recurrence kernels (diagnostic builds; not part of the product path)."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
variants = {"base": [], "pin_w": ["-DBABL_PIN_W"], "no_mm": ["-DBABL_NO_MM"], "no_p1load": ["-DBABL_NO_P1LOAD"],
            "no_store": ["-DBABL_NO_STORE"], "no_p1load_no_store": ["-DBABL_NO_P1LOAD", "-DBABL_NO_STORE"],
            "pin_w_no_p1": ["-DBABL_PIN_W", "-DBABL_NO_P1LOAD", "-DBABL_NO_STORE"]}
libs = {}
for name, fl in variants.items():
    so = "/tmp/abl_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *fl,
                           os.path.join(CS, "lstm.hip"), os.path.join(CS, "lstm_wide.hip"), os.path.join(CS, "lstm_split.hip"), os.path.join(CS, "api.hip"), "-o", so])
    libs[name] = ctypes.CDLL(so)
B, T, K, N = 32, 401, 34, 196
H, Hp = 2 * N, 416
M = B * T * K
dev = "cuda"
gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
whh = (torch.randn(2 * 400 * 4 * Hp, device=dev) * 0.05).to(torch.bfloat16)
hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
c = torch.empty(M, 2 * H, device=dev)
whhT = (torch.randn(2 * 400 * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
dh = torch.randn(M, 800, device=dev).to(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
def fwd(lib, path, rt):
    if path == "time": a = (B * K, T, K, T * K, K)
    else: a = (B * T, K, 1, K, 1)
    return lib.urse_lstm_bidir_fwd(P(gx.data_ptr()), ctypes.c_int64(8 * H), P(whh.data_ptr()), P(hout.data_ptr()), ctypes.c_int64(800),
        P(c.data_ptr()), H, Hp, a[0], a[1], ctypes.c_int64(a[2]), ctypes.c_int64(a[3]), ctypes.c_int64(a[4]), 1, 1, rt, None, P(st))
def bwd(lib, path, rt):
    if path == "time": a = (B * K, T, K, T * K, K)
    else: a = (B * T, K, 1, K, 1)
    return lib.urse_lstm_bidir_bwd(P(dh.data_ptr()), ctypes.c_int64(800), P(gx.data_ptr()), ctypes.c_int64(8 * H), P(c.data_ptr()),
        P(whhT.data_ptr()), H, a[0], a[1], ctypes.c_int64(a[2]), ctypes.c_int64(a[3]), ctypes.c_int64(a[4]), 1, rt, P(st))
whhb = (torch.randn(2 * 25 * 13 * 4 * 512, device=dev) * 0.05).to(torch.bfloat16)
def fwd_wide(lib, path, rt):
    if path == "time": a = (B * K, T, K, T * K, K)
    else: a = (B * T, K, 1, K, 1)
    return lib.urse_lstm_wide_fwd(P(gx.data_ptr()), ctypes.c_int64(8 * H), P(whhb.data_ptr()), P(hout.data_ptr()), ctypes.c_int64(800),
        P(c.data_ptr()), H, Hp, a[0], a[1], ctypes.c_int64(a[2]), ctypes.c_int64(a[3]), ctypes.c_int64(a[4]), 1, P(st))
plan = (ctypes.c_int64 * 4)()
assert libs["base"].urse_lstm_split_plan(H, B * K, 0, plan) == 0
print("split plan", list(plan))
xbuf = torch.empty(plan[2], device=dev, dtype=torch.float32)
errf = torch.zeros(1, device=dev, dtype=torch.int32)
def bwd_split(lib, path, rt):
    a = (B * K, T, K, T * K, K)
    return lib.urse_lstm_split_bwd(P(dh.data_ptr()), ctypes.c_int64(800), P(gx.data_ptr()), ctypes.c_int64(8 * H), P(c.data_ptr()),
        P(whhT.data_ptr()), P(xbuf.data_ptr()), P(errf.data_ptr()), H, a[0], a[1], ctypes.c_int64(a[2]), ctypes.c_int64(a[3]), ctypes.c_int64(a[4]), 0, P(st))
for fn, fname in ((bwd, "bwd"),):
    for path in ("time",):
        for rt in (0,):
            res = []
            for name, lib in libs.items():
                assert fn(lib, path, rt) == 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2): fn(lib, path, rt)
                torch.cuda.synchronize()
                res.append("%s %.2f" % (name, (time.perf_counter() - t0) / 2 * 1e3))
            print(fname, path, "rt", rt, " | ".join(res), "ms", flush=True)
