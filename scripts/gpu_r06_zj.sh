#!/bin/bash
# round 6, final tree: sustained 300 steps, rocprofv3 stats of the step, PMC HBM traffic of the fused cluster forward on both paths
export TMPDIR=/tmp
O=gpurun_out/r06zj; mkdir -p $O
python bench.py --gpus 1 --steps 300 --warmup 5 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $O/bench_300steps.json 2> $O/bench300.err; echo "sustained rc=$?"
python -c "import json; d=json.loads(open('$O/bench_300steps.json').read().strip().splitlines()[-1]); print('300 steps: %.2f ms/step, final loss' % d['ms_per_step'], d['final_loss'], 'peak HBM GB', d['peak_hbm_gb'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trainstep -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-160 && cp $f $O/trainstep_kernel_stats.csv
