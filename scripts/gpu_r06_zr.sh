#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06zr
python scripts/exp_clusterx_nt.py > gpurun_out/r06zr/exp_clusterx_nt.log 2>&1; tail -9 gpurun_out/r06zr/exp_clusterx_nt.log
