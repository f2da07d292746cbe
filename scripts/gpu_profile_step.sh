#!/bin/bash
# per-kernel times of the default train step: bash scripts/gpu_profile_step.sh <tag> [bench args, e.g. --dtype f16]
tag=${1:-step}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg --pretouch-gib 0 "$@" > $O/bench.log 2>&1; echo "rc=$?"
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 $O/bench.log | cut -c1-300
