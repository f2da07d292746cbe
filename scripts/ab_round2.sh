#!/bin/bash
# which of the round's changes carries the round-over-round difference: the previous round's tree, this tree under the previous round's switches, this tree
old=$1
run() {
  echo -n "$1: "
  (cd $2 && env $3 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode $(grep -q no-dist-leg bench.py && echo --no-dist-leg) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})")
}
R=$(pwd)
OLDCFG="URSE_NSPLIT_HELPERS=3 URSE_TN224_DEPTH=2 URSE_DEFER_MASKDEC_WGRADS=1 URSE_LSTM_CLUSTERX=0"
run "old tree            " $old "X=1"
run "new tree, old config" $R "$OLDCFG"
run "new tree            " $R "X=1"
run "new tree            " $R "X=1"
run "new tree, old config" $R "$OLDCFG"
run "old tree            " $old "X=1"
