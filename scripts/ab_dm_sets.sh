for rep in 1 2; do
for set in "STATIC" "-" "URSE_FFT_CONV_MIN_TAPS=100000000" "URSE_TN_SHADOW_WGS=120" "URSE_TN_NO_224=1 URSE_TN_SHADOW_WGS=120 URSE_FFT_CONV_MIN_TAPS=100000000"; do
  echo -n "[$set] "
  dm="--dynamic-mix"; envs="$set"
  if [ "$set" = "STATIC" ]; then dm=""; envs=""; fi
  if [ "$set" = "-" ]; then envs=""; fi
  env $envs python bench.py $dm --steps 10 --warmup 4 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernels_ms_per_step'].items()})"
done; done
