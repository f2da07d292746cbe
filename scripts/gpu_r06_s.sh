#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or fused_projection" > gpurun_out/r06s/test_cx.log 2>&1
echo "cx tests rc=$?"; tail -3 gpurun_out/r06s/test_cx.log
timeout 1200 python scripts/abl_clusterx.py D:XSTAMP=3 D:XSTAMP=3+D:XHORDER=0 D:XSTAMP=3+D:XPIPE=0 NO_CELL+NO_REC NO_HSTORE > gpurun_out/r06s/abl_clusterx_horder.log 2>&1
echo rc=$?; cat gpurun_out/r06s/abl_clusterx_horder.log
timeout 1500 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_xp1h0.so" "URSE_LIB_PATH=variants/liburse_xp0h1.so" "URSE_LIB_PATH=variants/liburse_xp0h0.so" > gpurun_out/r06s/ab_xpipe_horder.log 2>&1
cat gpurun_out/r06s/ab_xpipe_horder.log
