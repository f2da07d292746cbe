"""Row-wave LSTM forward (csrc/lstm_rw.hip) against the wide streaming kernel: bit equality of h / c / saved gates and time per launch
at the C2 band-path shape, plus ragged / strided sequence maps."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urgent2026_challenge_track1_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
N = 196
H, Hp = 2 * N, 416
torch.manual_seed(0)
whh = (torch.randn(2 * 4 * H, H, device=dev) * 0.05)
whhb = torch.empty(2 * 25 * 13 * 4 * 512, device=dev, dtype=torch.bfloat16)
ops.call("lstm_pack_blocks", whh, whhb, H, Hp, ops.stream_ptr())
whhb_rw = torch.empty_like(whhb)
ops.call("lstm_pack_blocks_rw", whh, whhb_rw, H, Hp, ops.stream_ptr())

def run(kind, gx, sm, save=True, tw=0):
    g = gx.clone()
    if kind == "wide":
        h, c = ops.lstm_fwd_wide(g, whhb, H, Hp, save=save, **sm)
    elif kind == "rw":
        h, c = ops.lstm_fwd_rw(g, whhb, H, Hp, save=save, target_wgs=tw, **sm)
    else:
        h, c = ops.lstm_fwd_rw(g, whhb_rw, H, Hp, save=save, target_wgs=tw, paired=True, **sm)
    torch.cuda.synchronize()
    return g, h, c

def eq(a, b):
    if a is None or b is None:
        return a is b
    return bool(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b))

cases = [("band ragged", dict(n_seq=100, seq_len=5, inner=1, outer=5, stride=1), 100 * 5),
         ("band 1 seq", dict(n_seq=1, seq_len=3, inner=1, outer=3, stride=1), 3),
         ("time map", dict(n_seq=2 * 34, seq_len=9, inner=34, outer=9 * 34, stride=34), 2 * 9 * 34),
         ("band 2000", dict(n_seq=2000, seq_len=34, inner=1, outer=34, stride=1), 2000 * 34)]
for name, sm, M in cases:
    gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
    for save in (True, False):
        a = run("wide", gx, sm, save)
        for kind in ("rw", "rw2"):
            b = run(kind, gx, sm, save)
            print("%-12s %-3s save=%d  gates %s  h %s  c %s" % (name, kind, save, eq(a[0], b[0]), eq(a[1], b[1]), eq(a[2], b[2])), flush=True)

B, T, K = 32, 401, 34
M = B * T * K
sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
a = run("wide", gx, sm)
for kind in ("rw", "rw2"):
    b = run(kind, gx, sm)
    print("C2 band path %s: gates %s  h %s  c %s" % (kind, eq(a[0], b[0]), eq(a[1], b[1]), eq(a[2], b[2])), flush=True)
    if not eq(a[1], b[1]):
        d = (a[1].float() - b[1].float()).abs()
        print("  max |dh| %.3e, mismatching rows %d" % (d.max().item(), int((d.amax(1) > 0).sum())))
    del b
del a
g = gx.clone()
for kind, tw in (("wide", 0), ("rw", 0), ("rw2", 0), ("rw2", 232), ("wide", 0), ("rw", 0), ("rw2", 0)):
    ts = []
    for _ in range(4):
        g.copy_(gx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if kind == "wide":
            ops.lstm_fwd_wide(g, whhb, H, Hp, save=True, **sm)
        elif kind == "rw":
            ops.lstm_fwd_rw(g, whhb, H, Hp, save=True, target_wgs=tw, **sm)
        else:
            ops.lstm_fwd_rw(g, whhb_rw, H, Hp, save=True, target_wgs=tw, paired=True, **sm)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-5s target_wgs %3d: %.3f ms (min of 4: %s)" % (kind, tw, min(ts), " ".join("%.3f" % v for v in ts)), flush=True)
