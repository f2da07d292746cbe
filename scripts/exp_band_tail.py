"""Band-path BPTT (12,832 sequences x 34 steps per direction): one launch of 802 32-sequence workgroups (3.13 rounds on 256 CUs)
against two CONCURRENT launches - 768 32-sequence workgroups and the last 544 sequences per direction as 68 16-sequence workgroups on a
second stream."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "urgent2026_challenge_track1_amd", "liburse_hip.so"))
B, T, K, N = 32, 401, 34, 196
H = 2 * N
M = B * T * K
dev = "cuda"
torch.manual_seed(0)
g0 = torch.rand(M, 8 * H, device=dev).to(torch.bfloat16)
c = torch.randn(M, 2 * H, device=dev)
whhT = (torch.randn(2 * 400 * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
dh = (0.1 * torch.randn(M, 800, device=dev)).to(torch.bfloat16)
P, L = ctypes.c_void_p, ctypes.c_int64
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n_seq = B * T
def launch(g, seq0, n, rows16, stream):
    off = seq0 * K
    rc = lib.urse_lstm_bidir_bwd(P(dh.data_ptr() + off * 800 * 2), L(800), P(g.data_ptr() + off * 8 * H * 2), L(8 * H), P(c.data_ptr() + off * 2 * H * 4),
                                 P(whhT.data_ptr()), H, n, K, L(1), L(K), L(1), 1, rows16, P(stream.cuda_stream))
    assert rc == 0, rc
def single(g):
    launch(g, 0, n_seq, 0, s1)
def split(g, n_main):
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    launch(g, 0, n_main, 0, s1)
    launch(g, n_main, n_seq - n_main, 1, s2)
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
ref = None
for name, fn in (("one launch (802 WGs)", single), ("768 + 68 concurrent", lambda g: split(g, 384 * 32)), ("736 + 132 concurrent", lambda g: split(g, 368 * 32)),
                 ("one launch (802 WGs)", single)):
    g = g0.clone(); torch.cuda.synchronize()
    fn(g); torch.cuda.synchronize()
    if ref is None: ref = g.clone()
    same = bool(torch.equal(g.view(torch.int16), ref.view(torch.int16)))
    ts = []
    for _ in range(5):
        g.copy_(g0); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%-24s %.3f ms (min of 5: %s)  equal to the single launch: %s" % (name, min(ts), " ".join("%.3f" % v for v in ts), same), flush=True)
