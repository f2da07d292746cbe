#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zp; mkdir -p $O
( time python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-metrics --no-f32-mode --no-dist-leg ) > $O/bench_short.json 2> $O/bench.err; echo "rc=$?"; tail -4 $O/bench.err
python -c "import json; d=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['inference_forward'])"
