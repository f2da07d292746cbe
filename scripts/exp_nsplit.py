"""(round 6: the touch / wide / helper-wave forms are compiled only with -DURSE_EXPERIMENTS - bash scripts/build_variant.sh nsx lstm_nsplit "-DURSE_EXPERIMENTS", then URSE_LIB_PATH=variants/liburse_nsx.so; the three-member kernel lives in csrc/experiments/ with scripts/abl_nsplit3.py)
N-split BPTT (csrc/lstm_nsplit.hip) against the streaming BPTT on the time path: agreement of the gate gradients, time per launch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from urgent2026_challenge_track1_amd import ops
dev, dt = "cuda", torch.bfloat16
N, H = 196, 392
torch.manual_seed(0)
lstm = torch.nn.LSTM(N, H, batch_first=True, bidirectional=True)
cat = lambda a, b: torch.cat([a, b]).detach().to(dev).contiguous()
pk = ops.lstm_pack(cat(lstm.weight_ih_l0, lstm.weight_ih_l0_reverse), cat(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse),
                   cat(lstm.bias_ih_l0, lstm.bias_ih_l0_reverse), cat(lstm.bias_hh_l0, lstm.bias_hh_l0_reverse), N, H, dt)
def case(B, T, K, time_runs=0):
    M = B * T * K
    sm = dict(n_seq=B * K, seq_len=T, inner=K, outer=T * K, stride=K)
    xr = ops.pack2d(torch.randn(M, N, device=dev), M, pk["Np"], dt)
    gx = ops.gemm_nt(xr, pk["wih"], pk["bias"])
    hout, c = ops.lstm_fwd(gx, pk["whh"], H, pk["Hp"], **sm)
    dh = ops.pack2d(torch.randn(M, 2 * H, device=dev) * 0.1, M, hout.shape[1], dt)
    g1, g2 = gx.clone(), gx.clone()
    ops.lstm_bwd(dh, g1, c, pk["whhT"], H, rows16=1, **sm)
    assert ops.lstm_nsplit_plan(H, sm["n_seq"]) is not None
    os.environ["URSE_NSPLIT_HELPERS"] = "0"
    g3 = gx.clone()
    ops.lstm_bwd_nsplit(dh, g3, c, pk["whhT"], H, **sm)
    os.environ["URSE_NSPLIT_HELPERS"] = "3"
    _, err = ops.lstm_bwd_nsplit(dh, g2, c, pk["whhT"], H, **sm)
    torch.cuda.synchronize()
    print("  helper waves == no helpers: %s" % torch.equal(g2, g3), flush=True)
    d = (g1.float() - g2.float()).abs()
    scale = g1.float().abs().max().item()
    print("B%d T%d K%d: err flag %d, max |d| / scale %.2e, mean %.2e, finite %s" % (B, T, K, int(err.item()), d.max().item() / scale, d.mean().item() / scale,
                                                                                bool(torch.isfinite(g2.float()).all())), flush=True)
    os.environ["URSE_NSPLIT_HELPERS"] = "0"; os.environ["URSE_NSPLIT_WIDE"] = "1"
    g5 = gx.clone()
    _, err5 = ops.lstm_bwd_nsplit(dh, g5, c, pk["whhT"], H, **sm)
    torch.cuda.synchronize()
    os.environ["URSE_NSPLIT_WIDE"] = "0"
    print("  seven waves x two tiles: err flag %d, == 13-wave form bit for bit: %s, max |d| vs streaming / scale %.2e" % (
        int(err5.item()), torch.equal(g5, g3), (g1.float() - g5.float()).abs().max().item() / scale), flush=True)
    os.environ["URSE_NSPLIT_TOUCH"] = "1"
    g6 = gx.clone()
    _, err6 = ops.lstm_bwd_nsplit(dh, g6, c, pk["whhT"], H, **sm)
    torch.cuda.synchronize()
    os.environ["URSE_NSPLIT_TOUCH"] = "0"
    print("  with the touch wave: err flag %d, == 13-wave form bit for bit: %s" % (int(err6.item()), torch.equal(g6, g3)), flush=True)
    for name in (("stream16", "nsplit0", "touch", "wide") * 2 if time_runs else ()):
        ts = []
        for _ in range(time_runs):
            g2.copy_(gx); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "stream16":
                ops.lstm_bwd(dh, g2, c, pk["whhT"], H, rows16=1, **sm)
            elif name == "touch":
                os.environ["URSE_NSPLIT_HELPERS"] = "0"; os.environ["URSE_NSPLIT_TOUCH"] = "1"
                ops.lstm_bwd_nsplit(dh, g2, c, pk["whhT"], H, **sm)
                os.environ["URSE_NSPLIT_TOUCH"] = "0"
            elif name == "wide":
                os.environ["URSE_NSPLIT_HELPERS"] = "0"; os.environ["URSE_NSPLIT_WIDE"] = "1"
                ops.lstm_bwd_nsplit(dh, g2, c, pk["whhT"], H, **sm)
                os.environ["URSE_NSPLIT_WIDE"] = "0"
            else:
                os.environ["URSE_NSPLIT_HELPERS"] = name[-1]
                ops.lstm_bwd_nsplit(dh, g2, c, pk["whhT"], H, **sm)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print("  %-8s %.3f ms (min of %d: %s)" % (name, min(ts), time_runs, " ".join("%.3f" % v for v in ts)), flush=True)
case(1, 7, 34)
case(2, 21, 20)
case(3, 9, 34)
case(32, 401, 34, time_runs=4)
