#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zt; mkdir -p $O
timeout 1200 python -m pytest tests/test_c4_fullsize_gpu.py tests/test_lstm_gpu.py tests/test_flow_gpu.py -x -q -m gpu -k "cluster2 or c4 or flow or C4" > $O/test_c4.log 2>&1; echo "rc=$?"; tail -2 $O/test_c4.log
timeout 900 bash scripts/ab_flow_env.sh "-" 2>&1 | tee $O/flow.log
