for kb in base 19 21 25 base 19 21 25; do
# (needs variants/liburse_kb<N>.so built with: bash scripts/build_variant.sh kb<N> lstm "-DURSE_BWD_KB2_STG=<N>")
  if [ $kb = base ]; then unset URSE_LIB_PATH; else export URSE_LIB_PATH=$PWD/variants/liburse_kb$kb.so; fi
  echo -n "KB2=$kb: "; python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
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
hs = int(g.view(torch.int16).to(torch.int64).sum().item())
ts = []
for _ in range(5):
    g.copy_(gx); torch.cuda.synchronize(); t0 = time.perf_counter()
    ops.lstm_bwd(dh, g, c, pk["whhT"], H, **sm); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("hash", hs, "%.3f ms" % min(ts))
PY
done
