#!/bin/bash
# HBM traffic passes of one bench step (FETCH_SIZE and WRITE_SIZE in separate runs) + the per-width calibration streams.
# usage (GPU box, repo root): bash scripts/gpu_pmc_traffic.sh <tag> [nocal]
tag=${1:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg --pretouch-gib 0"
C="python3 $R/scripts/pmc_calibrate.py"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1; echo "write rc=$?"
if [ "${2:-cal}" = "cal" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -- $C > $O/cal_fetch.log 2>&1; echo "cal fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -- $C > $O/cal_write.log 2>&1; echo "cal write rc=$?"
fi
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
