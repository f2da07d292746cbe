#!/bin/bash
# in-step A/B of the 960-point STFT launch: bash scripts/ab_stft.sh "URSE_STFT960_PIPE=0" "URSE_STFT960_PP=1" ... ("-" = defaults); every set twice, second pass reversed.
# Prints the step and the STFT launch's HIP-event time inside the step (us) + its fraction of the cold-copy yardstick.
sets=("$@")
run() {
  set="$1"; echo -n "[$set] "
  if [ "$set" = "-" ]; then set=""; fi
  env $set python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-metrics --no-flow --no-f32-mode --no-dist-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['stft_roofline']
print('step %.2f ms | stft %.1f us in the step = %.3f of 8 TB/s, %.2f of the cold copy (%.1f us)' % (d['ms_per_step'], 73.95e6/r['achieved']/1e3, r['frac'], r.get('frac_of_cold_stream_reference',0), 73.95e6/r.get('cold_stream_reference_GBs',1)/1e3))"
}
for s in "${sets[@]}"; do run "$s"; done
for ((i=${#sets[@]}-1; i>=0; i--)); do run "${sets[$i]}"; done
