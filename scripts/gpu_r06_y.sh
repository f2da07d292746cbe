#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06y
timeout 900 python scripts/exp_band_clusterx.py > gpurun_out/r06y/exp_band_clusterx.log 2>&1
echo rc=$?; cat gpurun_out/r06y/exp_band_clusterx.log
