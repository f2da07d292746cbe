#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06t
URSE_LIB_PATH=variants/liburse_h2.so timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or fused_projection" > gpurun_out/r06t/test_cx_h2.log 2>&1
echo "cx tests (h2) rc=$?"; tail -3 gpurun_out/r06t/test_cx_h2.log
timeout 1200 python scripts/abl_clusterx.py D:XSTAMP=3+D:XHORDER=2 D:XSTAMP=3+D:XHORDER=2+D:XHSLEEP2=40 D:XSTAMP=3+D:XHORDER=2+D:XDW=1 > gpurun_out/r06t/abl_clusterx_h2.log 2>&1
echo rc=$?; cat gpurun_out/r06t/abl_clusterx_h2.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_h2.so" "URSE_LIB_PATH=variants/liburse_h2s16.so" "URSE_LIB_PATH=variants/liburse_h2s40.so" "URSE_LIB_PATH=variants/liburse_h2d1.so" "URSE_LIB_PATH=variants/liburse_xp0h0.so" > gpurun_out/r06t/ab_h2.log 2>&1
cat gpurun_out/r06t/ab_h2.log
