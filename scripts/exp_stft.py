"""Cold-cache timing of the 960-point STFT (and iSTFT) at C2: every timed launch follows a 2 GiB fill that evicts L2 / the Infinity
Cache, as the launch inside the train step finds them.  python scripts/exp_stft.py [lib ...]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "urgent2026_challenge_track1_amd", "liburse_hip.so")]
B, L = 32, 192000
T, F = L // 480 + 1, 481
dev = "cuda"
x = torch.randn(B, L, device=dev)
spec = torch.empty(B, T, F, 2, device=dev)
junk = torch.empty(1 << 29, device=dev, dtype=torch.float32)
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
nbytes = B * (L * 4 + T * F * 8)
for path in libs:
    lib = ctypes.CDLL(path)
    def run():
        assert lib.urse_stft_fwd(P(x.data_ptr()), P(0), P(spec.data_ptr()), B, L, 960, 480, 1, P(st)) == 0
    run(); torch.cuda.synchronize()
    cold, hot = [], []
    for it in range(12):
        junk.fill_(float(it))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        cold.append(a.elapsed_time(b) * 1e3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): run()
    b.record(); torch.cuda.synchronize()
    h = a.elapsed_time(b) / 50 * 1e3
    cold.sort()
    c = cold[len(cold) // 2]
    print("%-30s cold %.1f us (%.2f TB/s, %.0f %% of 8 TB/s; min %.1f)   back-to-back %.1f us" % (os.path.basename(path), c, nbytes / c / 1e6, 100 * nbytes / c / 1e6 / 8, cold[0], h), flush=True)
