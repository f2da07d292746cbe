"""Where the fused cluster forward differs from the two-kernel form (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from urgent2026_challenge_track1_amd import ops
N, H, dev, dtype = 196, 392, "cuda", torch.bfloat16
B, T, K = 32, 25, 34
torch.manual_seed(5)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dtype)
M = B * T * K
x = torch.randn(B, T, K, N)
sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
xr = ops.pack2d(x.reshape(M, N).to(dev), M, pk["Np"], dtype)
gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
h1, c1, e1 = ops.lstm_fwd_cluster(gx, pk["whhq"], H, pk["Hp"], **sm)
for xa in (1, 0):
    g2, h2, c2, e2 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], xcd_aware=xa, **sm)
    d = (h1.float() - h2.float()).abs()[:, :2 * H].reshape(B, T, K, 2, H)
    bad = (d > 2e-2).nonzero()
    print("xcd_aware", xa, "err", int(e2.item()), "bad elements", bad.shape[0], "max", d.max().item())
    if bad.shape[0]:
        seq = bad[:, 0] * K + bad[:, 2]
        print("  sequences", sorted(set(seq.tolist()))[:40], "...", len(set(seq.tolist())))
        print("  t", sorted(set(bad[:, 1].tolist())), "dir", sorted(set(bad[:, 3].tolist())))
        print("  units", sorted(set(bad[:, 4].tolist()))[:60], len(set(bad[:, 4].tolist())))
        first = bad[bad[:, 1] == bad[:, 1].min()] if 0 in set(bad[:, 3].tolist()) else bad
        print("  first step entries", first[:10].tolist())
g1 = gx.view(torch.bfloat16)
g2, h2, c2, e2 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], xcd_aware=0, **sm)
# element (b=1, t=0, k=13, dir 0, unit 24): row = (b*T + t)*K + k
for (b, t, k, d, u) in ((1, 0, 13, 0, 24), (1, 0, 13, 0, 26), (0, 0, 5, 0, 24)):
    r = (b * T + t) * K + k
    print("elem", (b, t, k, d, u), "seq", b * K + k, "gates two-kernel", g1[r, d * 4 * H + u * 4: d * 4 * H + u * 4 + 4].float().tolist(),
          "fused", g2[r, d * 4 * H + u * 4: d * 4 * H + u * 4 + 4].float().tolist(), "c", c1[r, d * H + u].item(), c2[r, d * H + u].item(),
          "h", h1[r, d * H + u].item(), h2[r, d * H + u].item())
for rep in range(3):
    g3, h3, c3, e3 = ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], save=False, xcd_aware=1, **sm)
    d = (h3.float() - h2.float()).abs()[:, :2 * H].reshape(B, T, K, 2, H)
    bad = (d > 0).nonzero()
    print("no-save vs save: differing", bad.shape[0], "max", d.max().item(), "| vs two-kernel max", (h3.float() - h1.float()).abs().max().item())
    if bad.shape[0]:
        seq = bad[:, 0] * K + bad[:, 2]
        print("  rows in cluster", sorted(set((seq % 64).tolist())), "t", sorted(set(bad[:, 1].tolist())), "dir", sorted(set(bad[:, 3].tolist())))
        print("  units mod 8", sorted(set((bad[:, 4] % 8).tolist())), "first", bad[:6].tolist())
        b0 = bad[0].tolist(); r = (b0[0] * T + b0[1]) * K + b0[2]
        print("  values", h3[r, b0[3] * H + b0[4]].item(), h2[r, b0[3] * H + b0[4]].item(), h1[r, b0[3] * H + b0[4]].item())
