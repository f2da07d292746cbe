"""print VGPR / spill / occupancy per kernel of one .hip file: python scripts/kres.py lstm.hip [filter]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "urgent2026_challenge_track1_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
                    "-c", src, "-o", "/tmp/kres.o"] + extra, capture_output=True, text=True)
cur = None
for ln in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:110]
        vals = {}
        continue
    m = re.search(r"remark:\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur:
        vals[m.group(1)] = int(m.group(2))
        if m.group(1).startswith("LDS") and flt in cur:
            print("%-110s vgpr %3d agpr %3d spill %3d scratch %4d occ %d" % (cur, vals.get("VGPRs", 0), vals.get("AGPRs", 0), vals.get("VGPRs Spill", 0),
                                                                       vals.get("ScratchSize [bytes/lane]", 0), vals.get("Occupancy [waves/SIMD]", 0)))
if r.returncode:
    print(r.stderr[-2000:])
