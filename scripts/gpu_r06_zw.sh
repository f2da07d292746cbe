#!/bin/bash
# round 6, final verification (4): the final tree (row-tile instances)
export TMPDIR=/tmp
O=gpurun_out/r06zw; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_20steps.json
bash scripts/ab_round.sh variants/r05tree 2>&1 | tee $O/ab_round.log
python scripts/time_inference.py > $O/time_inference.log 2>&1; tail -12 $O/time_inference.log
