#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
python -m pytest tests/test_gemm_gpu.py -m gpu -x -q -k "f16 or mixed" > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -2 $O/test_gemm.log
python scripts/stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
python scripts/exp_tn_mixed.py 2>&1 | grep -v amdgpu.ids | tee $O/exp_tn_mixed.log
run() { echo -n "[$1 $2] "; env $2 python bench.py --dtype $1 --steps 10 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()}, 'loss', d['final_loss'])"; }
( run bf16 X=0; run f16 X=0; run f16 URSE_TN_ACT_F16=0; run f16 URSE_TN_ACT_F16=0; run f16 X=0; run bf16 X=0 ) 2>&1 | tee $O/ab_f16_step.log
