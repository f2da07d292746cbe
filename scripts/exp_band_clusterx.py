"""The band path's forward at C2 (2 x 12,832 sequences x 34 steps) through the fused cluster forward in rounds against the fused row-wave kernel:
each alone, 5 launches after a warm-up, ms per launch (host clock around a synchronised launch; the step A/B decides)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from urgent2026_challenge_track1_amd import ops
N, H, dev = 196, 392, "cuda"
T, K = 401, 34
for B, dtype in ((32, torch.bfloat16), (32, torch.float16), (16, torch.bfloat16), (8, torch.bfloat16), (4, torch.bfloat16), (3, torch.bfloat16)):
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
    M = B * T * K
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dtype)
    sm = dict(n_seq=B * T, seq_len=K, inner=1, outer=K, stride=1)
    print("B", B, dtype, "plan", ops.lstm_clusterx_plan(H, pk["Hp"], sm["n_seq"]), "pays", ops.band_clusterx_pays(H, pk["Hp"], sm["n_seq"]))

    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), sorted(ts)[len(ts) // 2], r
    for save in (True, False):
        a = timed(lambda: ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=save, **sm))
        b = timed(lambda: ops.lstm_fwd_rwx(xr, pk["wx"], pk["bias"], N, H, pk["Hp"], save=save, **sm))
        dh = (a[2][1][:, :2 * H].float() - b[2][1][:, :2 * H].float()).abs()
        print("  save=%d: cluster in rounds min %.3f median %.3f ms | row-wave min %.3f median %.3f ms | h differs max %.2e mean %.2e, flag %d" % (
            save, a[0], a[1], b[0], b[1], dh.max().item(), dh.mean().item(), int(a[2][3].item())), flush=True)
