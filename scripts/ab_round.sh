#!/bin/bash
# round over round on ONE box: the driver's bench command on this tree and on a kept build of the previous round's tree (variants/<dir>, git-ignored),
# each twice, second repetition in reverse order.  usage: bash scripts/ab_round.sh variants/r04tree
old=$1
run() {
  echo -n "$1: "
  (cd $2 && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode $(grep -q no-dist-leg bench.py && echo --no-dist-leg) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), round(d['value'],1), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})")
}
R=$(pwd)
run new $R; run old $old; run old $old; run new $R
