#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests/test_lstm_gpu.py -m gpu -x -q -k "nsplit" > $O/test_nsplit.log 2>&1; echo "nsplit tests rc=$?"; tail -2 $O/test_nsplit.log
python -m pytest tests/test_c2_fullsize_gpu.py tests/test_c2_parity_gpu.py -m gpu -x -q > $O/test_c2.log 2>&1; echo "c2 tests rc=$?"; tail -2 $O/test_c2.log
bash scripts/ab_step_sets.sh "URSE_LIB_PATH=variants/liburse_nopfi.so" "-" 2>&1 | tee $O/ab_nsplit_pfi.log
bash scripts/ab_step_sets.sh "URSE_TN224_DEPTH=4" "-" 2>&1 | tee $O/ab_tn224_interleave_v2.log
