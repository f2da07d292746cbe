#!/bin/bash
# round 6, call P: the band path through the fused cluster forward in rounds - parity, then the launch alone and the step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out/r06p
timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "rounds or fused_projection" > gpurun_out/r06p/test_rounds.log 2>&1
echo "rounds tests rc=$?"; tail -3 gpurun_out/r06p/test_rounds.log
timeout 600 python scripts/exp_band_clusterx.py > gpurun_out/r06p/exp_band_clusterx.log 2>&1
echo "exp rc=$?"; tail -12 gpurun_out/r06p/exp_band_clusterx.log
timeout 1200 bash scripts/ab_step_sets.sh "URSE_LSTM_BAND_CLUSTERX=1" "URSE_LSTM_BAND_CLUSTERX=0" > gpurun_out/r06p/ab_band_clusterx.log 2>&1
cat gpurun_out/r06p/ab_band_clusterx.log
