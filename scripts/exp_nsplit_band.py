"""N-split BPTT on the BAND path (12,832 sequences x 34 steps: 802 pairs = 1,604 workgroups, more than the chip holds) against the 32-row streaming BPTT."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urgent2026_challenge_track1_amd import ops, _lib
lib = _lib.load()
dev, dt = "cuda", torch.bfloat16
N, H = 196, 392
torch.manual_seed(0)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
B, T, K = 32, 401, 34
M = B * T * K
sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dt)
gx, hout, c = ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], **sm)
dh = ops.pack2d(torch.randn(M, 2 * H, device=dev) * 0.1, M, hout.shape[1], dt)
flags = torch.zeros(16384, device=dev, dtype=torch.int32)
err = ops.kernel_error_flag(torch.device(dev, 0))
P, L = ctypes.c_void_p, ctypes.c_int64
def nsplit(g):
    rc = lib.urse_lstm_nsplit_bwd(P(dh.data_ptr()), L(dh.stride(0)), P(g.data_ptr()), L(g.stride(0)), P(c.data_ptr()), P(pk["whhT"].data_ptr()),
                                  P(flags.data_ptr()), P(err.data_ptr()), H, sm["n_seq"], sm["seq_len"], L(1), L(K), L(1), -1, P(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, (rc, lib.urse_last_error())
g1, g2 = gx.clone(), gx.clone()
ops.lstm_bwd(dh, g1, c, pk["whhT"], H, **sm)
nsplit(g2); torch.cuda.synchronize()
d = (g1.float() - g2.float()).abs(); scale = g1.float().abs().max().item()
print("band path: err flag %d, max |d| / scale %.2e, mean %.2e" % (int(err.item()), d.max().item() / scale, d.mean().item() / scale), flush=True)
for name in ("stream32", "nsplit", "stream32", "nsplit"):
    ts = []
    for _ in range(4):
        g2.copy_(gx); torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "stream32":
            ops.lstm_bwd(dh, g2, c, pk["whhT"], H, **sm)
        else:
            nsplit(g2)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("  %-8s %.3f ms (min of 4: %s)  err %d" % (name, min(ts), " ".join("%.3f" % v for v in ts), int(err.item())), flush=True)
