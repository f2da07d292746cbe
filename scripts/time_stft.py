import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd import ops
x = torch.randn(32, 192000, device="cuda")
s = ops.stft_forward(x, 960, 480); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): s = ops.stft_forward(x, 960, 480)
b.record(); torch.cuda.synchronize()
t1 = a.elapsed_time(b) / 20 * 1e3
a.record()
for _ in range(20): w = ops.istft_forward(s, 960, 480, 192000)
b.record(); torch.cuda.synchronize()
print("NF", os.environ.get("URSE_STFT_NF"), "stft %.1f us  istft %.1f us" % (t1, a.elapsed_time(b) / 20 * 1e3))
