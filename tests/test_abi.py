"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/urse.h declares, and rejects
bad arguments with an error code + message before touching the GPU."""
import ctypes

import pytest


def test_exports_every_declared_symbol(lib):
    from urgent2026_challenge_track1_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 25 and "urse_stft_fwd" in names and "urse_lstm_bidir_fwd" in names
    for n in names:
        assert hasattr(lib, n), n
    protos = _lib.prototypes()
    assert set(protos) == set(names)
    assert lib.urse_version() >= 1


def test_argument_errors_do_not_touch_the_gpu(lib):
    rc = lib.urse_stft_fwd(None, None, None, 1, 100, 64, 32, 1, None)
    assert rc == -1 and b"urse_stft_fwd" in lib.urse_last_error()
    rc = lib.urse_gemm_nt(None, 0, None, 0, None, 0, None, None, 0, 4, 4, 32, 1, 1, 0, None)
    assert rc == -1 and b"null" in lib.urse_last_error()
    rc = lib.urse_lstm_bidir_fwd(ctypes.c_void_p(16), 8, ctypes.c_void_p(16), ctypes.c_void_p(16), 8, None, 7, 32, 1, 1, 1,
                                 1, 1, 0, 1, 0, None, None)
    assert rc == -1


def test_shape_queries_refuse_what_the_kernels_cannot_run(lib):
    """the fused cluster forward is written for N <= 200 input channels (six slabs of W_ih + the folded seventh) and H = 392;
    the query used to accept N up to 224 (channels 200 .. 223 silently dropped) and any multiple of 56 below 392 (a gather that waits for
    chunks nobody publishes).  Round 6: the mixed-operand weight-gradient GEMMs answer for their own shapes."""
    assert lib.urse_lstm_clusterx_supported(196, 224, 392, 416) == 1
    assert lib.urse_lstm_clusterx_supported(200, 224, 392, 416) == 1
    assert lib.urse_lstm_clusterx_supported(208, 224, 392, 416) == 0
    assert lib.urse_lstm_clusterx_supported(196, 224, 336, 416) == 0
    assert lib.urse_lstm_clusterx_supported(196, 224, 392, 448) == 0
    assert lib.urse_gemm_tn_act_f16_supported(436288, 196, 784, 0, 1, 1, 0) == 1           # fc gradient at C2
    assert lib.urse_gemm_tn_act_f16_supported(436288, 1568, 196, 392, 1, 34, 401) == 1     # dual gradient, time path
    assert lib.urse_gemm_tn_act_f16_supported(436288, 1568, 196, 392, 1, 1, 34) == 1       # band path
    assert lib.urse_gemm_tn_act_f16_supported(20604, 1568, 196, 392, 1, 34, 101) == 0      # rows not whole 32-row stages
    assert lib.urse_gemm_tn_act_f16_supported(436288, 3072, 384, 768, 1, 1, 48) == 0       # H = 768: no 224-row tiles
    assert lib.urse_gemm_tn_act_f16_supported(436288, 196, 784, 0, 0, 1, 0) == 0           # the transposed instance carries the column sums


def test_grouped_gemm_validates_the_host_mirror_before_launching(lib):
    """urse_gemm_nt_grouped_h reads the host copy of the per-band records: a bad one is an error code, not a launch"""
    import numpy as np
    good = [4096, 8192, 12288, 0, 0, 64, 64, 64, 16, 32, 64, 0]       # {A, B, C, bias, resid, lda, ldb, ldc, M, N, K, ldr}
    bad_k = list(good); bad_k[10] = 40; bad_k[5] = bad_k[6] = 40       # K not a multiple of 32 (bf16)
    bad_ld = list(good); bad_ld[7] = 16                                 # ldc < N
    for rec in (bad_k, bad_ld):
        host = np.asarray([good, rec], dtype=np.int64)
        rc = lib.urse_gemm_nt_grouped_h(ctypes.c_void_p(4096), ctypes.c_void_p(host.ctypes.data), 2, 1, 1, 0, None)
        assert rc == -1 and b"urse_gemm_nt_grouped_h" in lib.urse_last_error()
    rc = lib.urse_gemm_nt_grouped_h(None, None, 2, 1, 1, 0, None)
    assert rc == -1


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    from urgent2026_challenge_track1_amd import ops
    from urgent2026_challenge_track1_amd._lib import UrseError
    with pytest.raises(UrseError):
        ops.stft_forward(torch.randn(1, 4000), 320, 160)
    with pytest.raises(UrseError):
        ops.mr_l1_loss(torch.randn(1, 4000), torch.randn(1, 4000))


def test_product_package_never_imports_the_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urgent2026_challenge_track1_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_product_package_calls_no_vendor_math_library():
    """GEMMs, FFTs, reductions are this repository's kernels: no torch.fft (rocFFT), torch.matmul / mm / bmm / einsum / conv (rocBLAS /
    hipBLASLt / MIOpen), torch.linalg on the product path; the C++ side links no rocBLAS / hipBLASLt / rocFFT / MIOpen."""
    import os
    import re
    import subprocess
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urgent2026_challenge_track1_amd")
    banned = re.compile(r"torch\.fft\b|torch\.(matmul|mm|bmm|einsum|addmm|baddbmm|conv1d|conv2d)\b|torch\.linalg\b|F\.(linear|conv1d|conv2d)\b|@\s*\w+\.T\b")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                for i, line in enumerate(open(os.path.join(dp, f)).read().splitlines()):
                    code = line.split("#")[0]
                    if '"""' in code or code.lstrip().startswith(("'", '"')):
                        continue
                    assert not banned.search(code), (f, i + 1, line.strip())
    so = os.path.join(root, "liburse_hip.so")
    if os.path.exists(so):
        needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
        for libname in ("rocblas", "hipblas", "rocfft", "hipfft", "MIOpen", "rccl"):
            assert libname not in needed, libname


def test_cluster_forward_plan_in_rounds(lib):
    """Round 6: urse_lstm_clusterx_plan = {C, clusters per direction, sequences per cluster, rows_pad, hx elements, counters, rounds}.  Up to clusters x 64
    sequences per direction it is urse_lstm_cluster_plan's geometry in one round; above, every cluster takes 64 per round.  (No GPU here: the library plans
    for 256 CUs; a reservation leaves fewer clusters and more rounds.)"""
    plan = (ctypes.c_int64 * 7)()
    one = (ctypes.c_int64 * 6)()
    assert lib.urse_lstm_clusterx_plan(392, 416, 1088, 0, plan) == 0 and lib.urse_lstm_cluster_plan(392, 416, 1088, 0, one) == 0
    assert list(plan)[:4] == list(one)[:4] and plan[5] == one[5] and plan[6] == 1 and plan[0] == 7 and plan[1] == 18
    assert lib.urse_lstm_clusterx_plan(392, 416, 1152, 0, plan) == 0 and plan[6] == 1
    assert lib.urse_lstm_cluster_plan(392, 416, 1153, 0, one) != 0                           # the one-round plan refuses ...
    assert lib.urse_lstm_clusterx_plan(392, 416, 1153, 0, plan) == 0 and list(plan)[1:4] == [18, 64, 64] and plan[6] == 2      # ... this one takes a second round
    assert lib.urse_lstm_clusterx_plan(392, 416, 12832, 0, plan) == 0 and plan[6] == 12      # the band path at C2
    hx = ctypes.c_int64()
    assert lib.urse_lstm_clusterx_hx_elems(392, 416, 12832, 0, ctypes.byref(hx)) == 0 and hx.value == plan[4] == 2 * 2 * 18 * 64 * 432
    assert lib.urse_lstm_clusterx_plan(392, 416, 12832, 32, plan) == 0 and plan[1] == 15 and plan[6] == 14      # beside 32 reserved CUs
    assert lib.urse_lstm_clusterx_plan(392, 416, 12832, 250, plan) != 0                      # no room for a single cluster
    assert lib.urse_lstm_clusterx_plan(768, 768, 100, 0, plan) != 0                          # not this kernel's shape
