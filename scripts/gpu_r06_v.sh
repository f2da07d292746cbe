#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06v
timeout 1500 python scripts/abl_clusterx.py D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=3 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=4 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=3+D:XAD=6 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=4+D:XAD=6 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=0+D:XAD=6 D:XSTAMP=3+D:XHORDER=0+D:XSTAMP_W=4+D:XAD=4 > gpurun_out/r06v/abl_clusterx_xad.log 2>&1
echo rc=$?; cat gpurun_out/r06v/abl_clusterx_xad.log
