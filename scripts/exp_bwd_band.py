"""Band-path BPTT (12,832 x 34): the shipped 32-row / 8-wave kernel (variant 0) against variants of URSE_BWD_VARIANT
(7: two-slot pipelined input loads; 8: 48 rows on four waves, one per SIMD - round 5).  python scripts/exp_bwd_band.py [variants...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from urgent2026_challenge_track1_amd import ops
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
g = gx.clone()
ops.lstm_bwd(dh, g, c, pk["whhT"], H, **sm); torch.cuda.synchronize()
print("checksum %%.6f" %% g.float().abs().double().sum().item(), "hash", int(g.view(torch.int16).to(torch.int64).sum().item()))

ts = []
for _ in range(5):
    g.copy_(gx); torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.lstm_bwd(dh, g, c, pk["whhT"], H, **sm); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("%%.3f ms (min of 5: %%s)" %% (min(ts), " ".join("%%.3f" %% v for v in ts)))
''' % ROOT
for v in (sys.argv[1:] or ["0", "8", "0", "8"]):
    env = dict(os.environ, URSE_BWD_STAGED_STORES="1", URSE_BWD_VARIANT=v)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("variant=%s" % v, r.stdout.strip().replace("\n", " | "), r.stderr[-300:] if r.returncode else "", flush=True)

