#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zx; mkdir -p $O
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_xm6.so" "URSE_LIB_PATH=variants/liburse_xm4.so" "URSE_LIB_PATH=variants/liburse_xmE.so" "URSE_LIB_PATH=variants/liburse_xm1.so" > $O/ab_midpoll.log 2>&1
cat $O/ab_midpoll.log
