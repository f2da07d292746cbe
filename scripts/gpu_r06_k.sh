#!/bin/bash
# round 6 verification: whole GPU suite, smoke, the driver's bench command, kernel stats of the step, HBM traffic passes, the flow leg's profile
O=gpurun_out/r06k; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/gputest.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench_20steps.json
bash scripts/gpu_profile_step.sh r06k_prof
bash scripts/gpu_pmc_traffic.sh r06k_pmc nocal
bash scripts/gpu_profile_flow.sh r06k_flow
