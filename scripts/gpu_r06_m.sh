#!/bin/bash
O=gpurun_out/r06m; mkdir -p $O
timeout 600 python -m pytest tests/test_lstm_gpu.py -m gpu -x -q -k "role_split or 32_row" > $O/test_roles.log 2>&1; echo "roles tests rc=$?"; tail -5 $O/test_roles.log
timeout 900 python -m pytest tests/test_c2_fullsize_gpu.py tests/test_c2_parity_gpu.py -m gpu -x -q > $O/test_c2.log 2>&1; echo "c2 tests rc=$?"; tail -2 $O/test_c2.log
bash scripts/ab_step_sets.sh "URSE_BWD_ROLES=0" "-" 2>&1 | tee $O/ab_bwd_roles.log
