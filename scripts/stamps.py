"""In-kernel cycle stamps (s_memtime) of the three kernels that had none: the dual weight-gradient GEMM, the band path's BPTT and
the band path's fused forward - each built alone with its stamp switch, run alone at the C2 shape, median shader-clock cycles per phase.
python scripts/stamps.py [tn224] [bwd] [rwx]      (diagnostic builds in /tmp; the shipping library is not touched)"""
import ctypes, os, subprocess, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "csrc")
which = sys.argv[1:] or ["tn224", "bwd", "rwx"]
P, L = ctypes.c_void_p, ctypes.c_int64
B, T, K, N = 32, 401, 34, 196
H, Hp, Np = 2 * N, 416, 224
M = B * T * K
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream


def build(name, files, flags):
    so = "/tmp/stamp_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", *flags,
                           *[os.path.join(CS, f) for f in files], "-o", so])
    return ctypes.CDLL(so)


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


def med(a):
    return float(np.median(a))


if "tn224" in which:
    g = torch.Generator().manual_seed(0)
    A = (0.1 * torch.randn(M, 4 * H, generator=g)).bfloat16().to(dev)
    X = torch.zeros(M, Np, dtype=torch.bfloat16); X[:, :N] = torch.randn(M, N, generator=g).bfloat16(); X = X.to(dev)
    Hh = torch.zeros(M, Hp, dtype=torch.bfloat16); Hh[:, :H] = torch.tanh(torch.randn(M, H, generator=g)).bfloat16(); Hh = Hh.to(dev)
    c1, c2, cs = torch.zeros(4 * H, N, device=dev), torch.zeros(4 * H, H, device=dev), torch.zeros(4 * H, device=dev)
    for tag, flags in (("stamps", ["-DT224STAMP=3"]), ("stamps + a wait behind the fragment reads", ["-DT224STAMP=3", "-DT224STAMP_SPLIT"])):
        lib = build("tn224_%d" % len(flags), ["gemm.hip", "norm.hip", "api.hip"], flags)
        base = build("tn224_base", ["gemm.hip", "norm.hip", "api.hip"], [])
        for wgs in (256, 112, 84):
            def run(l=lib):
                rc = l.urse_gemm_tn_dual(P(A.data_ptr()), L(4 * H), P(X.data_ptr()), L(Np), P(c1.data_ptr()), L(N), P(cs.data_ptr()), P(Hh.data_ptr()), L(Hp),
                                         P(c2.data_ptr()), L(H), L(M), L(4 * H), L(N), L(H), L(-K), L(K), L(T), L(0), L(H), 1, wgs, P(st))
                assert rc == 0, rc
            ms, ms0 = timed(run), timed(lambda: run(base))
            buf = np.zeros(512 * 8, dtype=np.uint64)
            assert lib.urse_diag_tn224_stamps(buf.ctypes.data_as(P)) == 0
            s = buf.reshape(512, 8).astype(np.int64)
            ok = (s[:, 0] > 0) & (s[:, 5] > 0)
            s = s[ok][4:-2]
            nx = np.roll(s[:, 0], -1)[:-1]
            it = nx - s[:-1, 0]
            print("gemm_tn_dual224_kernel<2>, %s, target %d workgroups: %.3f ms (unstamped build %.3f ms); wave 0 of workgroup 3, %d iterations of two 32-row stages, median cycles:" % (tag, wgs, ms, ms0, len(s)))
            print("   own DMAs landed (s_waitcnt vmcnt(0))   %7.0f" % med(s[:, 1] - s[:, 0]))
            print("   barrier                                %7.0f" % med(s[:, 2] - s[:, 1]))
            print("   issue of the next two stages' DMAs     %7.0f" % med(s[:, 3] - s[:, 2]))
            if len(flags) > 1:
                print("   stage a: fragment reads                %7.0f" % med(s[:, 6] - s[:, 3]))
                print("   stage a: 35 MFMAs                      %7.0f" % med(s[:, 4] - s[:, 6]))
            else:
                print("   stage a: fragment reads + 35 MFMAs     %7.0f" % med(s[:, 4] - s[:, 3]))
            print("   stage b: fragment reads + 35 MFMAs     %7.0f" % med(s[:, 5] - s[:, 4]))
            print("   iteration                              %7.0f   (= %.2f us per 32-row stage at the launch's rate: %.2f us)" % (
                med(it), med(it) / 2 / 2.1e3, ms * 1e3 / (M / 32 / max(1, (wgs // 14)))), flush=True)

if "bwd" in which:
    gx = torch.randn(M, 8 * H, device=dev).to(torch.bfloat16)
    c = torch.randn(M, 2 * H, device=dev)
    whhT = (torch.randn(2 * 400 * 4 * H, device=dev) * 0.05).to(torch.bfloat16)
    dh = torch.randn(M, 800, device=dev).to(torch.bfloat16)
    files = ["lstm.hip", "lstm_wide.hip", "lstm_split.hip", "api.hip"]
    lib, base = build("bwd", files, ["-DBWSTAMP=5"]), build("bwd_base", files, [])
    def run(l=lib):
        rc = l.urse_lstm_bidir_bwd(P(dh.data_ptr()), L(800), P(gx.data_ptr()), L(8 * H), P(c.data_ptr()), P(whhT.data_ptr()), H, B * T, K, L(1), L(K), L(1), 1, 0, P(st))
        assert rc == 0, rc
    ms, ms0 = timed(run), timed(lambda: run(base))
    buf = np.zeros(512 * 16, dtype=np.uint64)
    assert lib.urse_diag_bwd_stamps(buf.ctypes.data_as(P)) == 0
    s = buf.reshape(512, 16)[2:K - 2].astype(np.int64)
    nx = buf.reshape(512, 16)[3:K - 1].astype(np.int64)
    print("band-path BPTT (lstm_bwd_kernel<bf16, 32 rows, 8 waves, staged stores>): %.3f ms (unstamped build %.3f ms); wave 0 of workgroup 5, median cycles per step:" % (ms, ms0))
    for ui in range(4):
        print("   cell phase, unit tile %d (inputs loaded -> gradients in the LDS tile)   %7.0f" % (ui, med(s[:, 1 + ui] - s[:, ui])))
    print("   arrival at the barrier -> behind it                                   %7.0f" % med(s[:, 6] - s[:, 5]))
    print("   staged stores of the step's gate gradients issued                     %7.0f" % med(s[:, 7] - s[:, 6]))
    prev = 7
    for ui in range(4):
        print("   recurrent product, unit tile %d (49 fragments of W_hh^T, 98 MFMAs)     %7.0f" % (ui, med(s[:, 8 + ui] - s[:, prev])))
        prev = 8 + ui
    print("   step                                                                  %7.0f   (cell phase %.0f, barrier %.0f, stores %.0f, products %.0f)" % (
        med(nx[:, 0] - s[:, 0]), med(s[:, 4] - s[:, 0]), med(s[:, 6] - s[:, 4]), med(s[:, 7] - s[:, 6]), med(s[:, 11] - s[:, 7])), flush=True)

if "rwx" in which:
    xn = torch.randn(M, Np, device=dev).to(torch.bfloat16)
    wx = (torch.randn(2 * 25 * 20 * 4 * 512, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(8 * H, device=dev)
    gates = torch.empty(M, 8 * H, device=dev, dtype=torch.bfloat16)
    hout = torch.zeros(M, 800, device=dev, dtype=torch.bfloat16)
    c = torch.zeros(M, 2 * H, device=dev)
    lib, base = build("rwx", ["lstm_rwx.hip", "api.hip"], ["-DRXSTAMP=6"]), build("rwx_base", ["lstm_rwx.hip", "api.hip"], [])
    def run(l=lib):
        rc = l.urse_lstm_rwx_fwd(P(xn.data_ptr()), L(Np), P(wx.data_ptr()), P(bias.data_ptr()), P(gates.data_ptr()), L(8 * H), P(hout.data_ptr()), L(800),
                                 P(c.data_ptr()), N, Np, H, Hp, B * T, K, L(1), L(K), L(1), 1, 0, 1, None, P(st))
        assert rc == 0, rc
    ms, ms0 = timed(run), timed(lambda: run(base))
    buf = np.zeros(64 * 4, dtype=np.uint64)
    assert lib.urse_diag_rwx_stamps(buf.ctypes.data_as(P)) == 0
    s = buf.reshape(64, 4)[2:K - 1].astype(np.int64)
    tot = s[:, 0] + s[:, 1] + s[:, 2]
    print("band-path fused forward (lstm_fwd_rwx_kernel): %.3f ms (unstamped build %.3f ms); the LOADER wave of workgroup 6, median cycles per time step (200 ring stages of 10 KB):" % (ms, ms0))
    print("   issuing the stage's ten LDS-DMAs                 %8.0f  (%.0f per stage)" % (med(s[:, 0]), med(s[:, 0]) / 200))
    print("   waiting for stage k + 2 to LAND (s_waitcnt)      %8.0f  (%.0f per stage)" % (med(s[:, 1]), med(s[:, 1]) / 200))
    print("   waiting at the stage's BARRIER for the compute waves %4.0f  (%.0f per stage)" % (med(s[:, 2]), med(s[:, 2]) / 200))
    print("   step                                             %8.0f  = %.1f us" % (med(tot), med(tot) / 2.1e3), flush=True)
