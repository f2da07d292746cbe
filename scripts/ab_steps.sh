#!/bin/bash
# same box: the train step over 6 / 20 / 100 timed steps, and 100 with blocking uploads
B="python bench.py --warmup 5 --no-flow --no-f32-mode --no-dist-leg --no-metrics --no-cpu-baseline"
for rep in 1 2; do
for n in 6 20 100; do echo -n "steps $n: "; $B --steps $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"; done
echo -n "steps 100 pageable: "; URSE_PAGEABLE_UPLOADS=1 $B --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
done
