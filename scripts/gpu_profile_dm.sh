#!/bin/bash
# per-kernel times of the --dynamic-mix bench step (device simulator inside the step)
# usage (GPU box, repo root): bash scripts/gpu_profile_dm.sh <tag>
tag=${1:-dm}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --dynamic-mix --steps 4 --warmup 2 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $O/bench.log 2>&1; echo "rc=$?"
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
head -40 $O/kernel_stats.csv
