#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zd; mkdir -p $O
timeout 1200 python scripts/abl_clusterx.py D:XSTAMP=3+D:XSTAMP_W=4 D:XSTAMP=3+D:XSTAMP_W=4+D:XHEAD=1 D:XSTAMP=3+D:XSTAMP_W=4+D:XHEAD=1+D:XFETCH_POS=1 D:XSTAMP=3+D:XSTAMP_W=4+D:XPRIO=1 D:XSTAMP=3+D:XSTAMP_W=0+D:XPRIO=1 D:XSTAMP=3+D:XSTAMP_W=4+D:XPRIO=2 D:XSTAMP=3+D:XSTAMP_W=0+D:XPRIO=2 > $O/abl_clusterx_head_prio.log 2>&1
echo rc=$?; grep -v "^   \(gathered\|at b1\|after b2\|published\|issued\)" $O/abl_clusterx_head_prio.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_hd1.so" "URSE_LIB_PATH=variants/liburse_hd1fp1.so" "URSE_LIB_PATH=variants/liburse_pr1.so" "URSE_LIB_PATH=variants/liburse_pr2.so" "URSE_LIB_PATH=variants/liburse_hd1pr1.so" > $O/ab_head_prio.log 2>&1
cat $O/ab_head_prio.log
