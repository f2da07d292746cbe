"""What binds the 960-point STFT launch when it starts from cold caches (as the first kernel of a train step does)?  Every timed launch follows a
2 GiB fill; one HIP-event pair per launch (the bench's method).  Yardsticks with the same method: an empty launch, a float4 copy of 37 + 37 MB,
a pass with the STFT's own traffic shape (24.6 MB read, 49.4 MB written).  python scripts/exp_stft_cold.py [lib ...]   (env URSE_STFT960_* apply)"""
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


def timed(fn, cold=True, n=12):
    ts = []
    for it in range(n):
        if cold:
            junk.fill_(float(it))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def b2b(fn, n=50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


if os.environ.get("EXP_YARDSTICKS", "1") == "1":
    z = torch.empty(1, device=dev)
    n = 37 * 1000 * 1000 // 4
    ca, cb = torch.randn(n, device=dev), torch.empty(n, device=dev)
    xs = x.view(1, -1)
    o2 = spec.view(-1)[:2 * x.numel()].view(2, -1)
    for name, fn in (("empty launch", lambda: z.fill_(1.0)), ("copy 37 MB -> 37 MB", lambda: torch.mul(ca, 2.0, out=cb)),
                     ("read 24.6 MB -> write 49.2 MB", lambda: torch.mul(xs.expand(2, -1), 2.0, out=o2))):
        fn(); torch.cuda.synchronize()
        c, cmin = timed(fn)
        w, wmin = timed(fn, cold=False)
        print("%-34s cold %.1f us (min %.1f)   warm, own event pair %.1f   back-to-back %.1f" % (name, c, cmin, w, b2b(fn)), flush=True)
for path in libs:
    lib = ctypes.CDLL(path)
    def run():
        assert lib.urse_stft_fwd(P(x.data_ptr()), P(0), P(spec.data_ptr()), B, L, 960, 480, 1, P(st)) == 0
    run(); torch.cuda.synchronize()
    c, cmin = timed(run)
    w, wmin = timed(run, cold=False)
    tag = " ".join("%s=%s" % (k[13:], v) for k, v in os.environ.items() if k.startswith("URSE_STFT960_"))
    print("%-34s cold %.1f us (min %.1f; %.2f TB/s = %.0f %% of 8 TB/s)   warm, own event pair %.1f   back-to-back %.1f"
          % ((os.path.basename(path) + " " + tag)[:34], c, cmin, nbytes / c / 1e6, 100 * nbytes / c / 1e6 / 8, w, b2b(run)), flush=True)
