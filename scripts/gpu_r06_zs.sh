#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zs; mkdir -p $O
timeout 1200 python -m pytest tests/test_c4_fullsize_gpu.py tests/test_lstm_gpu.py tests/test_flow_gpu.py -x -q -m gpu -k "cluster2 or c4 or flow or C4" > $O/test_c4.log 2>&1; echo "rc=$?"; tail -3 $O/test_c4.log
timeout 1500 bash scripts/ab_flow_env.sh "-" "URSE_LIB_PATH=variants/liburse_c2sync3.so" "URSE_LIB_PATH=variants/liburse_c2sync3.so" "-" 2>&1 | tee $O/ab_c2_sync3.log
