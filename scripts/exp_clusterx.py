"""Time path forward at C2 (1,088 sequences x 401 steps): gate GEMM + cluster forward (two kernels) against the cluster forward with the projection
fused (csrc/lstm_clusterx.hip).  One process, interleaved rounds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
dev = "cuda"
N, H, B, T, K = 196, 392, 32, 401, 34
M = B * T * K
sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
for dt in (torch.bfloat16, torch.float16):
    torch.manual_seed(0)
    lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
    cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
    pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                       cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dt)
    def two():
        gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
        return ops.lstm_fwd_cluster(gx, pk["whhq"], H, pk["Hp"], **sm) + (gx,)
    def one():
        return ops.lstm_fwd_clusterx(xr, pk["wihq"], pk["whhq"], pk["bias"], N, H, pk["Hp"], **sm)
    h1, c1, e1, gx = two(); g2, h2, c2, e2 = one(); torch.cuda.synchronize()
    dh = (h1.float() - h2.float()).abs()
    print(dt, "err flags", int(e1.item()), int(e2.item()), "| h max %.2e mean %.2e | c max %.2e | gates max %.2e" %
          (dh.max().item(), dh.mean().item(), (c1 - c2).abs().max().item(), (gx.view(torch.bfloat16).float() - g2.float()).abs().max().item()), flush=True)
    del h1, c1, gx, g2, h2, c2
    res = {"two": [], "one": []}
    for rnd in range(4):
        for name, f in (("two", two), ("one", one)):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) * 1e3); del r
    print(dt, "gate GEMM + cluster forward: %s ms | fused: %s ms" % (" ".join("%.3f" % v for v in res["two"]), " ".join("%.3f" % v for v in res["one"])), flush=True)
