#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zn; mkdir -p $O
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or row_tiles or fused_projection" > $O/test_cx.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_cx.log
python scripts/time_inference.py > $O/time_inference.log 2>&1; tail -12 $O/time_inference.log
