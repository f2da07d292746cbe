#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zb; mkdir -p $O
for rep in 1 2; do
for set in "URSE_LSTM_TIME_CLUSTERX_ROUNDS=1" "URSE_LSTM_TIME_CLUSTERX_ROUNDS=0"; do
  echo -n "[B 48, $set] "
  env $set python bench.py --batch 48 --steps 5 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],1), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
done; done 2>&1 | tee $O/ab_b48.log
tail -3 $O/err.log
