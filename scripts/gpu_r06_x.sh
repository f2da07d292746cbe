#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06x
URSE_LIB_PATH=variants/liburse_a30fp1.so timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or fused_projection" > gpurun_out/r06x/test_cx.log 2>&1
echo "cx tests (a30fp1) rc=$?"; tail -3 gpurun_out/r06x/test_cx.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_a30.so" "URSE_LIB_PATH=variants/liburse_a40.so" "URSE_LIB_PATH=variants/liburse_a31.so" "URSE_LIB_PATH=variants/liburse_fp1.so" "URSE_LIB_PATH=variants/liburse_a30fp1.so" "URSE_LIB_PATH=variants/liburse_xp0h0.so" > gpurun_out/r06x/ab_dma_split.log 2>&1
cat gpurun_out/r06x/ab_dma_split.log
