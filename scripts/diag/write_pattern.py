"""HBM write rate: GEMM output tiles vs linear fill (diagnostic)."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/write_pattern.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", os.path.join(here, "write_pattern.hip"), "-o", so])
lib = ctypes.CDLL(so)
M, pitch = 32 * 401 * 34, 6272
buf = torch.empty(M * pitch, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(name, seg, mode, nt):
    f = lambda: lib.run(ctypes.c_void_p(buf.data_ptr()), ctypes.c_long(M), ctypes.c_long(pitch), seg, mode, nt, ctypes.c_void_p(st))
    assert f() == 0; torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%-34s %.3f ms  %.2f TB/s" % (name, ms, M * pitch / ms / 1e9), flush=True)
t("linear", 448, 1, 0); t("linear nt", 448, 1, 1)
for seg in (448, 896, 3136, 6272):
    t("tiles 256 x %d B" % seg, seg, 0, 0); t("tiles 256 x %d B nt" % seg, seg, 0, 1)
