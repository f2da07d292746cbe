#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06r
timeout 1200 python scripts/abl_clusterx.py base D:XPIPE=0 D:XSTAMP=3 D:XSTAMP=3+D:XPIPE=0 NO_CELL NO_REC NO_CELL+NO_REC NO_PROJ NO_AREAD NO_GATHER+NO_XSTORE > gpurun_out/r06r/abl_clusterx_xpipe.log 2>&1
echo rc=$?; cat gpurun_out/r06r/abl_clusterx_xpipe.log
