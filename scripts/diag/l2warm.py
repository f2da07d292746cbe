"""Diagnostic: L2-hit stream of a CU beside HBM misses of other waves of the same CU; scalar / vector line-touch rates (see l2warm.hip)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/l2warm.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", os.path.join(here, "l2warm.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = "cuda"
HOT = 1254400 // 1024 * 1024                      # one direction's W_hh
hot = torch.randn(HOT // 4, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
P = lambda t: ctypes.c_void_p(t.data_ptr())

def run(name, grid, waves, n_stream, mode, passes, cold_per_wave):
    ncold = max(1, waves - n_stream)
    cold = torch.empty(grid * ncold * cold_per_wave + 4096, dtype=torch.uint8, device=dev)
    stamps = torch.zeros(grid * waves * 2, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc = lib.run_mix(P(hot), ctypes.c_long(HOT), P(cold), ctypes.c_long(cold_per_wave), passes, n_stream, mode, waves, grid, P(stamps), P(sink), ctypes.c_void_p(st))
        assert rc == 0
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(grid, waves, 2).astype(np.float64) * 10e-9    # 100 MHz ticks -> seconds
    out = "%-46s" % name
    if n_stream:
        d = (s[:, :n_stream, 1].max(1) - s[:, :n_stream, 0].min(1))
        out += " hot: %7.1f us/pass %6.1f GB/s/CU" % (d.mean() / passes * 1e6, HOT * passes / d.mean() / 1e9)
    if mode and waves > n_stream:
        d = (s[:, n_stream:, 1].max(1) - s[:, n_stream:, 0].min(1))
        nb = cold_per_wave * ncold
        out += " | cold: %7.1f us, %6.2f GB/s/CU of lines (%.0f lines/us), whole chip %.0f GB/s" % (
            d.mean() * 1e6, nb / d.mean() / 1e9, nb / 128 / d.mean() / 1e6, nb * grid / d.max() / 1e9 if False else nb * grid / d.mean() / 1e9)
    print(out, flush=True)
    del cold

G = int(os.environ.get("GRID", "136"))
run("16 streamers", G, 16, 16, 0, 200, 0)
run("12 streamers", G, 16, 12, 0, 200, 0)
run("12 streamers + 4 vector HBM streams", G, 16, 12, 1, 200, 32 << 20)
run("12 streamers + 4 vector touchers", G, 16, 12, 2, 200, 32 << 20)
run("12 streamers + 4 scalar touchers", G, 16, 12, 3, 200, 4 << 20)
run("15 streamers + 1 vector HBM stream", G, 16, 15, 1, 200, 64 << 20)
run("15 streamers + 1 scalar toucher", G, 16, 15, 3, 200, 4 << 20)
for w in (1, 4, 16):
    run("%d vector HBM streams alone" % w, G, w, 0, 1, 0, 16 << 20)
    run("%d vector touchers alone" % w, G, w, 0, 2, 0, 16 << 20)
    run("%d scalar touchers alone" % w, G, w, 0, 3, 0, 2 << 20)
# helpers on their own CUs: 16 workgroups (2 per XCD) of 16 waves
run("16 WGs x 16 scalar touchers", 16, 16, 0, 3, 0, 2 << 20)
run("16 WGs x 16 vector touchers", 16, 16, 0, 2, 0, 16 << 20)

# scalar touches, then a vector read of the same 88 KB: does a touch every 128 (64) bytes make the read an L2 hit?
def warm(name, grid, tstride, wait_ticks=0, rounds=64, nbytes=90112, region=1 << 20):
    cold = torch.empty(grid * rounds * region + 4096, dtype=torch.uint8, device=dev)
    stamps = torch.zeros(grid, dtype=torch.int64, device=dev)
    rc = lib.run_warm(P(cold), ctypes.c_long(region), nbytes, tstride, rounds, wait_ticks, grid, P(stamps), P(sink), ctypes.c_void_p(st))
    assert rc == 0
    torch.cuda.synchronize()
    d = stamps.cpu().numpy().astype(np.float64) * 10e-9 / rounds
    print("%-46s read of 88 KB: %6.2f us mean, %6.2f max" % (name, d.mean() * 1e6, d.max() * 1e6), flush=True)
    del cold
for G2 in (136,):
    warm("no touch (HBM)", G2, 0)
    warm("touch every 128 B", G2, 128)
    warm("touch every 64 B", G2, 64)
    warm("touch every 128 B, read 5 us later", G2, 128, 500)
    warm("touch every 128 B, read 20 us later", G2, 128, 2000)
    warm("no touch, second read of the same region", G2, 0, rounds=64, region=0)
