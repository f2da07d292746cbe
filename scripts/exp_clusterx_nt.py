"""Step time of the fused cluster forward by instance (row tiles per step): the time path (T = 401) at batch sizes that select NT = 1 .. 4, each alone,
with and without saving; us per step = ms per launch / 401."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
N, H, dev, dtype = 196, 392, "cuda", torch.bfloat16
T, K = 401, 34
torch.manual_seed(0)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
for B in (1, 4, 7, 12, 14, 20, 21, 32):
    M = B * T * K
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    plan = ops.lstm_clusterx_plan(H, pk["Hp"], sm["n_seq"])
    bound = -(-sm["n_seq"] // (plan[1] - 2))
    nt = 4 if bound > 48 else 3 if bound > 32 else 2 if bound > 16 else 1
    out = []
    for save in (True, False):
        for env in (None, "4"):
            if env: os.environ["URSE_CLUSTERX_NT"] = env
            else: os.environ.pop("URSE_CLUSTERX_NT", None)
            f = lambda: ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=save, **sm)
            f(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            out.append(min(ts))
    os.environ.pop("URSE_CLUSTERX_NT", None)
    print("B %2d: %4d sequences, %2d rows per cluster (bound %2d) -> NT %d: save %.3f ms = %.2f us/step (full instance %.2f) | forward only %.3f ms = %.2f us/step (full %.2f)" % (
        B, sm["n_seq"], plan[2], bound, nt, out[0], out[0] * 1e3 / T, out[1] * 1e3 / T, out[2], out[2] * 1e3 / T, out[3] * 1e3 / T), flush=True)
