#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zm; mkdir -p $O
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or row_tiles" > $O/test_cx.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_cx.log
timeout 1500 bash scripts/ab_step_sets.sh "-" "URSE_LSTM_CLUSTERX_TAIL=0" "URSE_LSTM_CLUSTERX_TAIL=0" "-" > $O/ab_tail.log 2>&1; cat $O/ab_tail.log
python scripts/time_inference.py > $O/time_inference.log 2>&1; grep "B=32\|B=16\|operands" $O/time_inference.log
URSE_LSTM_CLUSTERX_TAIL=0 python scripts/time_inference.py > $O/time_inference_notail.log 2>&1; grep "B=32\|B=16\|operands" $O/time_inference_notail.log
