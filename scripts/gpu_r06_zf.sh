#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zf; mkdir -p $O
URSE_LIB_PATH=variants/liburse_pf32.so timeout 900 python -m pytest tests/test_lstm_gpu.py -q -m gpu -k "rounds or fused_projection" -s 2>&1 | grep -v "^$" | tail -30 > $O/test_cx_pf32.log; cat $O/test_cx_pf32.log | cut -c1-220
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_pf32.so" "URSE_LIB_PATH=variants/liburse_pf32.so" "-" > $O/ab_proj_f32.log 2>&1
cat $O/ab_proj_f32.log
