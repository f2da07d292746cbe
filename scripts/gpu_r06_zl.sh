#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zl; mkdir -p $O
timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_c2_parity_gpu.py tests/test_f16_gpu.py tests/test_train_gpu.py -x -q -m gpu > $O/test_cx.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_cx.log
python scripts/dbg_nt.py 40 2>&1 | tail -4 | cut -c1-200
python scripts/time_inference.py > $O/time_inference_nt.log 2>&1; tail -12 $O/time_inference_nt.log
