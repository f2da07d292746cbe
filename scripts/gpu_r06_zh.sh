#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zh; mkdir -p $O
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds" > $O/test_rounds.log 2>&1; echo "rc=$?"; tail -5 $O/test_rounds.log
