"""HBM yardstick: achievable fill / copy / read rates for a gate-matrix sized buffer (diagnostic)."""
import time, torch
n = 32 * 401 * 34 * 3136
a = torch.empty(n, device="cuda", dtype=torch.bfloat16); b = torch.empty_like(a)
def t(name, fn, nbytes, k=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    print("%-10s %7.3f ms %6.2f TB/s" % (name, dt * 1e3, nbytes / dt / 1e12), flush=True)
t("fill", lambda: a.zero_(), 2 * n)
t("copy", lambda: b.copy_(a), 4 * n)
t("read(sum)", lambda: a.view(torch.int16).sum(), 2 * n)
t("read(max)", lambda: a.amax(), 2 * n)
