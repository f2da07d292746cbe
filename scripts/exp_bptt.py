"""Time-path / band-path BPTT variants at the C2 shape: time per launch and bit-equality of the outputs with variant 0.
python scripts/exp_bptt.py [lib ...]   (libs: paths of variant builds; default the in-tree library)
URSE_BWD_VARIANT is switched per run (0 default, 1 = 16 waves + cross-step prefetch, 2 = 8 waves + prefetch, 3 = in-step single round trip)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or [os.path.join(ROOT, "urgent2026_challenge_track1_amd", "liburse_hip.so")]
B, T, K, N = 32, 401, 34, 196
H = 2 * N
M = B * T * K
dev = "cuda"
torch.manual_seed(0)
g0 = torch.rand(M, 8 * H, device=dev).to(torch.bfloat16)            # saved gate activations in (0, 1)
c = torch.randn(M, 2 * H, device=dev)
whhT = (torch.randn(2 * 400 * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
dh = (0.1 * torch.randn(M, 800, device=dev)).to(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
P, L = ctypes.c_void_p, ctypes.c_int64
paths = {"time": (B * K, T, K, T * K, K), "band": (B * T, K, 1, K, 1)}
ref = {}
for lib_path in libs:
    lib = ctypes.CDLL(lib_path)
    for path in (os.environ.get("EXP_PATHS", "time").split(",")):
        a = paths[path]
        for var in os.environ.get("EXP_VARIANTS", "0,1,2,3").split(","):
            os.environ["URSE_BWD_VARIANT"] = var
            g = g0.clone()
            def run():
                rc = lib.urse_lstm_bidir_bwd(P(dh.data_ptr()), L(800), P(g.data_ptr()), L(8 * H), P(c.data_ptr()), P(whhT.data_ptr()), H,
                                             a[0], a[1], L(a[2]), L(a[3]), L(a[4]), 1, 0, P(st))
                assert rc == 0, rc
            run()
            torch.cuda.synchronize()
            out = g.clone()
            key = path
            same = None
            if key in ref:
                same = bool(torch.equal(out.view(torch.int16), ref[key].view(torch.int16)))
                if not same:
                    d = (out.float() - ref[key].float()).abs()
                    bad = (out.view(torch.int16) != ref[key].view(torch.int16)).nonzero()
                    r0, c0 = bad[0].tolist()
                    print("   differs in %d of %d elements, max |d| %.3e (ref max %.3e); first at row %d (b %d, t %d, k %d) col %d (dir %d, unit %d, gate %d): %r vs %r"
                          % (bad.shape[0], out.numel(), float(d.max()), float(ref[key].float().abs().max()), r0, r0 // (T * K), (r0 // K) % T, r0 % K,
                             c0, c0 // (4 * H), (c0 % (4 * H)) // 4, c0 % 4, float(out[r0, c0]), float(ref[key][r0, c0])))
                    ts_ = sorted(set(((bad[:, 0] // K) % T).tolist()))
                    print("   time steps with differences: %s ... %s (%d distinct)" % (ts_[:6], ts_[-6:], len(ts_)))
            else:
                ref[key] = out
            ts = []
            for _ in range(3):
                g.copy_(g0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%-28s %s variant %s: %.3f ms (min of 3; %s)  bit-equal to the first: %s"
                  % (os.path.basename(lib_path), path, var, min(ts), " ".join("%.3f" % v for v in ts), same), flush=True)
            del g, out
