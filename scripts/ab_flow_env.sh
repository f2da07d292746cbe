#!/bin/bash
# same-box A/B of the BSRNN-Flow leg under environment settings: bash scripts/ab_flow_env.sh "A=1" "-" ...
for rep in 1 2; do
  for set in "$@"; do
    echo -n "[$set] "
    if [ "$set" = "-" ]; then set=""; fi
    env $set python bench.py --model flow --steps 4 --pretouch-gib 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['flow_c4']; print(round(d['train_ms_per_step'],2), round(d['enhance_ms'],1))"
  done
done
