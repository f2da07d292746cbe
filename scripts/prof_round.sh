#!/bin/bash
# per-kernel times of this tree and of a kept build of the previous round's tree on ONE box: bash scripts/prof_round.sh variants/r04tree <tag>
old=$1; tag=${2:-profround}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O/new $O/old
cd /tmp
for w in new old; do
  T=$R; [ $w = old ] && T=$R/$old
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/prof -- python3 $T/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg --pretouch-gib 0 > $O/$w/bench.log 2>&1; echo "$w rc=$?"
  find $O/$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${w}_kernel_stats.csv
  find $O/$w -name "*.db" -delete 2>/dev/null; find $O/$w -name "*kernel_trace.csv" -delete 2>/dev/null; rm -rf $O/$w/prof
done
