#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
( time python -m pytest tests/test_train_gpu.py -m gpu -x -q ) > $O/gputest_tail.log 2>&1; echo "train gpu tests rc=$?"; tail -4 $O/gputest_tail.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -o trainstep -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160 && cp $f $O/trainstep_kernel_stats.csv
