"""Forward-only latency / throughput of BSRNN_SE (N = 196, 6 layers, 48 kHz) at a few batch sizes.  Diagnostic."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd.bsrnn import BSRNN_SE
from urgent2026_challenge_track1_amd import ops
for dtype in (torch.bfloat16, torch.float16):
  m = BSRNN_SE(num_channel=196, num_layer=6, compute_dtype=dtype).cuda().eval()
  print("operands", dtype)
  for B in (1, 4, 16, 32):
      x = 0.1 * torch.randn(B, 192000, device="cuda")
      lens = torch.full((B,), 192000)
      with torch.no_grad():
          m(x, lens, 48000); torch.cuda.synchronize()
          t0 = time.perf_counter()
          for _ in range(3): y = m(x, lens, 48000)
          torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / 3
      print("B=%d x 4 s @ 48 kHz: %.1f ms per forward, %.1f utt/s, real-time factor %.5f" % (B, dt * 1e3, B / dt, dt / (4.0 * B)), flush=True)

  ops.poll_kernel_errors(torch.device("cuda", 0), sync=True)
  print("  kernels:", {k: v for k, v in ops.launch_counts(reset=True).items() if v and k.startswith("lstm")})
