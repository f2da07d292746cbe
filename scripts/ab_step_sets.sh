#!/bin/bash
# same-box A/B of the train step under several environment settings: bash scripts/ab_step_sets.sh "A=1 B=2" "A=3" ... ("-" = defaults),
# every set twice, interleaved
for rep in 1 2; do
  for set in "$@"; do
    echo -n "[$set] "
    if [ "$set" = "-" ]; then set=""; fi
    env $set python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
  done
done
