#!/bin/bash
# SQ / cache counter passes of one bench step (rocprofv3 --pmc, each pass its own run, --kernel-trace only).
# usage (GPU box, from the repo root): bash scripts/gpu_pmc_sq.sh <tag>
tag=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-metrics --no-dist-leg --pretouch-gib 0"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
run c TCC_HIT_sum TCC_MISS_sum
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run e GRBM_GUI_ACTIVE
# keep only the csv files (the merged-back directory is capped)
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
