#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zk; mkdir -p $O
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "row_tiles or rounds or fused_projection" > $O/test_cx.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_cx.log
python scripts/time_inference.py > $O/time_inference_nt.log 2>&1; tail -12 $O/time_inference_nt.log
URSE_CLUSTERX_NT=4 python scripts/time_inference.py > $O/time_inference_nt4.log 2>&1; tail -12 $O/time_inference_nt4.log
timeout 1200 bash scripts/ab_step_sets.sh "-" "URSE_CLUSTERX_NT=4" > $O/ab_nt.log 2>&1; cat $O/ab_nt.log
