"""C2 time path forward: gate-projection GEMM + cluster recurrence vs the fused kernel (HIP events, isolated)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
N, B, T, K = 196, 32, 401, 34
H, dt, dev = 2 * N, torch.bfloat16, "cuda"
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
M = B * T * K
xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dt)
sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def unfused():
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    return ops.lstm_fwd_cluster(gx, pk["whhq"], H, pk["Hp"], **sm)
def fused():
    return ops.lstm_fwd_cluster_x(xr, pk["wihq"], pk["bias"], pk["whhq"], pk["Np"], H, pk["Hp"], **sm)
def gemm_only():
    return ops.gemm_nt(xr, pk["wih"], pk["bias"])
print("gate projection GEMM %.3f ms | GEMM + cluster fwd %.3f ms | fused cluster fwd %.3f ms" % (t(gemm_only), t(unfused), t(fused)))
