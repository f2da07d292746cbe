#!/bin/bash
# round 6, call Q: phase 2 of the fused cluster forward interleaved by hand (XPIPE) - parity, then the step A/B against the compiler's order
export TMPDIR=/tmp
mkdir -p gpurun_out/r06q
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or fused_projection" > gpurun_out/r06q/test_cx.log 2>&1
echo "cx tests rc=$?"; tail -3 gpurun_out/r06q/test_cx.log
timeout 1200 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_xpipe0.so" > gpurun_out/r06q/ab_xpipe.log 2>&1
cat gpurun_out/r06q/ab_xpipe.log
