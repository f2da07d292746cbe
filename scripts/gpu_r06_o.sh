#!/bin/bash
O=gpurun_out/r06o; mkdir -p $O
python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "default_bench_line" > $O/test_leg.log 2>&1; echo "leg test rc=$?"; tail -2 $O/test_leg.log
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $O/bench_300steps.json 2>/dev/null; echo "sustained rc=$?"; python -c "
import json; d=json.load(open('$O/bench_300steps.json')); print('300 steps:', round(d['ms_per_step'],2), 'ms/step, final loss', d['final_loss'], 'peak HBM GB', round(d['peak_hbm_gb'],1))"
python bench.py --dynamic-mix --steps 10 --warmup 4 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $O/bench_dynamic_mix.json 2>/dev/null; echo "dynamic mix rc=$?"; python -c "
import json; d=json.load(open('$O/bench_dynamic_mix.json')); print('dynamic mix:', round(d['ms_per_step'],2), 'ms/step', d.get('dynamic_mix'))"
python scripts/time_inference.py 2>&1 | grep -v amdgpu.ids | tee $O/time_inference.log
