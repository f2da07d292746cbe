#!/bin/bash
# same-box A/B of the train step under settings of one environment variable: bash scripts/ab_step_env.sh VAR v1 v2 ... (each twice, interleaved)
var=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo -n "$var=$v: "
    env $var=$v python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
  done
done
