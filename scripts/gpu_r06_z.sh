#!/bin/bash
# round 6, verification of the tree with the band path in rounds + the hand-interleaved phase 2: whole GPU suite, smoke, the driver's bench command,
# the same step under rocprofv3, inference timings, A/B against the previous kernel (variants/liburse_xp0h0.so = compiler's order) and the row-wave band forward
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_20steps.json
python scripts/time_inference.py > $O/time_inference.log 2>&1; tail -12 $O/time_inference.log
bash scripts/ab_step_sets.sh "-" "URSE_LSTM_BAND_CLUSTERX=0" "URSE_LIB_PATH=variants/liburse_xp0h0.so" "URSE_LIB_PATH=variants/liburse_xp0h0.so URSE_LSTM_BAND_CLUSTERX=0" > $O/ab_round6_forward.log 2>&1; cat $O/ab_round6_forward.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o trainstep -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160 && cp $f $O/trainstep_kernel_stats.csv
