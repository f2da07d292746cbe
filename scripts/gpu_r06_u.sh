#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06u
timeout 1500 python scripts/abl_clusterx.py D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=1 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=3 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=4 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=6 D:XSTAMP=3+D:XHORDER=0+D:XDW=3 D:XSTAMP=3+D:XHORDER=0+D:XDW=4 D:XSTAMP=3+D:XHORDER=1+D:XDW=4 > gpurun_out/r06u/abl_clusterx_waves.log 2>&1
echo rc=$?; cat gpurun_out/r06u/abl_clusterx_waves.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_xp1h0.so" "URSE_LIB_PATH=variants/liburse_xp1h0d3.so" "URSE_LIB_PATH=variants/liburse_xp1h1d3.so" "URSE_LIB_PATH=variants/liburse_xp1h0d4.so" > gpurun_out/r06u/ab_xdw.log 2>&1
cat gpurun_out/r06u/ab_xdw.log
