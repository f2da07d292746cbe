#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "f16 or mixed" > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 $O/test_gemm.log
python -m pytest tests/test_c4_fullsize_gpu.py -m gpu -q -s -k "enhance" > $O/test_c4.log 2>&1; echo "c4 enhance rc=$?"; grep "C4 enhance" $O/test_c4.log | cut -c1-400; tail -2 $O/test_c4.log
python -m pytest tests/test_train_gpu.py tests/test_entry_gpu.py tests/test_flow_gpu.py -m gpu -x -q -s -k "inference or flow" > $O/test_inf.log 2>&1; echo "inference/flow rc=$?"; grep "trained checkpoint" $O/test_inf.log; tail -2 $O/test_inf.log
bash scripts/gpu_profile_step.sh r06e_prof_bf16 --dtype bf16
bash scripts/gpu_profile_step.sh r06e_prof_f16 --dtype f16
