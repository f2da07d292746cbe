"""Small shapes of the fused cluster forward (partial clusters): error flag and difference to the two-kernel form (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from urgent2026_challenge_track1_amd import ops
N, H, dev, dtype = 196, 392, "cuda", torch.bfloat16
torch.manual_seed(5)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
for (B, T, K) in ((2, 9, 20), (3, 7, 34), (2, 40, 34), (4, 12, 34), (32, 5, 34)):
    M = B * T * K
    x = torch.randn(B, T, K, N)
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    h1, c1, e1 = ops.lstm_fwd_cluster(gx, pk["whhq"], H, pk["Hp"], **sm)
    torch.cuda.synchronize()
    print((B, T, K), "plan", ops.lstm_cluster_plan(H, pk["Hp"], sm["n_seq"]), "two-kernel err", int(e1.item()), flush=True)
    e1.zero_()
    for xa in (0, 1):
        g2, h2, c2, e2 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], xcd_aware=xa, **sm)
        torch.cuda.synchronize()
        d = (h1.float() - h2.float()).abs()[:, :2 * H].reshape(B, T, K, 2, H)
        print("   xcd_aware", xa, "err", int(e2.item()), "max diff", d.max().item(), "per t", [round(d[:, t].max().item(), 3) for t in range(min(T, 8))], flush=True)
        e2.zero_()
