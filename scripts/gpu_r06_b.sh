#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_stft_gpu.py tests/test_loss_gpu.py -m gpu -x -q > $O/test_stft.log 2>&1; echo "stft tests rc=$?"; tail -3 $O/test_stft.log
python -m pytest tests/test_bsrnn_gpu.py -m gpu -x -q > $O/test_bsrnn.log 2>&1; echo "bsrnn tests rc=$?"; tail -3 $O/test_bsrnn.log
bash scripts/ab_stft.sh "URSE_STFT960_PIPE=0" "URSE_STFT960_PP=1" "URSE_STFT960_PP=2" "URSE_STFT960_PP=3" "URSE_STFT960_PP=4" 2>&1 | tee $O/ab_stft.log
