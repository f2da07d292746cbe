#!/bin/bash
# round 6, final verification: whole GPU suite, smoke, the driver's bench command, rocprofv3 stats of the step, round-over-round A/B against the round-5 tree
export TMPDIR=/tmp
O=gpurun_out/r06ze; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_20steps.json
bash scripts/ab_round.sh variants/r05tree 2>&1 | tee $O/ab_round.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trainstep -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-160 && cp $f $O/trainstep_kernel_stats.csv
