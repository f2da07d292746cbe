#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_gemm_gpu.py -m gpu -x -q > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -3 $O/test_gemm.log
python -m pytest tests/test_f16_gpu.py -m gpu -x -q -s > $O/test_f16.log 2>&1; echo "f16 tests rc=$?"; grep "f16 C2 kernel set" $O/test_f16.log; tail -3 $O/test_f16.log
python -m pytest tests/test_c2_fullsize_gpu.py -m gpu -x -q -k "reproducible or f16" > $O/test_full.log 2>&1; echo "fullsize rc=$?"; tail -3 $O/test_full.log
# f16-forward step with the mixed-operand weight gradients vs the two-copy form vs bf16, same box, both orders
run() { echo -n "[$1 $2] "; env $2 python bench.py --dtype $1 --steps 10 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()}, 'loss', d['final_loss'])"; }
( run bf16 X=0; run f16 URSE_TN_ACT_F16=1; run f16 URSE_TN_ACT_F16=0; run f16 URSE_TN_ACT_F16=0; run f16 URSE_TN_ACT_F16=1; run bf16 X=0 ) 2>&1 | tee $O/ab_f16_step.log
