#!/bin/bash
O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_lstm_gpu.py -m gpu -x -q > $O/test_lstm.log 2>&1; echo "lstm tests rc=$?"; tail -2 $O/test_lstm.log
python -m pytest tests/test_c2_fullsize_gpu.py tests/test_c2_parity_gpu.py tests/test_bsrnn_gpu.py -m gpu -x -q > $O/test_c2.log 2>&1; echo "c2 tests rc=$?"; tail -2 $O/test_c2.log
python scripts/stamps.py bwd 2>&1 | grep -v amdgpu.ids | tee $O/stamps_bwd_pairs.log
bash scripts/ab_step_sets.sh "URSE_BWD_PAIRS=0" "-" 2>&1 | tee $O/ab_bwd_pairs.log
