#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zu; mkdir -p $O
timeout 1500 bash scripts/ab_flow_env.sh "-" "URSE_LIB_PATH=variants/liburse_c2always.so" "URSE_LIB_PATH=variants/liburse_c2always.so" "-" 2>&1 | tee $O/ab_c2_sync3_cond.log
