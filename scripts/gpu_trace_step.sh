#!/bin/bash
# dispatch sequence of one train step (queues, gaps): bash scripts/gpu_trace_step.sh <tag> [bench args]
tag=${1:-trace}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg --pretouch-gib 0 "$@" > $O/bench.log 2>&1; echo "rc=$?"
f=$(find $O -name "*kernel_trace.csv" | head -1)
head -1 $f > $O/trace_header.txt
python3 $R/scripts/trace_step.py $f $O/step_sequence.txt; tail -40 $O/step_sequence.txt
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -delete 2>/dev/null
