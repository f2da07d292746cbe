"""Known-byte-count streams for calibrating FETCH_SIZE / WRITE_SIZE per access width (run under rocprofv3 --pmc ...):
reads then writes 2 GiB (larger than the 256 MiB Infinity Cache) with 4-, 8- and 16-byte per-lane accesses."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from urgent2026_challenge_track1_amd._lib import call, stream_ptr
n = 2 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
sink = torch.zeros(4, device="cuda")
buf.zero_()
torch.cuda.synchronize()
for write in (0, 1):
    for width in (4, 8, 16):
        call("diag_stream", buf, sink, n, width, write, stream_ptr())
        torch.cuda.synchronize()
print("ok")
