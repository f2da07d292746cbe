#!/bin/bash
# same-box A/B of the train step under settings of one environment variable: bash scripts/ab_step_env.sh VAR v1 v2 ...
# Each setting twice; the second repetition runs the settings in REVERSE order: on this pool a run that follows another one is ~1 ms faster than the
# first run of a call, so a fixed order credits that millisecond to whatever comes later (found at the end of round 4: profiles/r04_ab_order_bias_v1.log).
var=$1; shift
vals=("$@")
run() {
  echo -n "$var=$1: "
  env $var=$1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
}
for v in "${vals[@]}"; do run $v; done
for ((i=${#vals[@]}-1; i>=0; i--)); do run ${vals[$i]}; done
