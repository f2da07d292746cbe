#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "tn" > $O/test_gemm.log 2>&1; echo "gemm tn tests rc=$?"; tail -2 $O/test_gemm.log
URSE_TN224_DEPTH=4 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "tn_dual or poison" > $O/test_gemm4.log 2>&1; echo "gemm tn tests (depth 4) rc=$?"; tail -2 $O/test_gemm4.log
python scripts/exp_tn224_depth.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_tn224_interleave.log
bash scripts/ab_step_sets.sh "URSE_TN224_DEPTH=4" "-" 2>&1 | tee $O/ab_tn224_interleave.log
