#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r06w
timeout 1500 python scripts/abl_clusterx.py D:XHORDER=0 D:XHORDER=0+D:XPIPE=0 NO_DMA+D:XHORDER=0 NO_DMA+D:XHORDER=0+D:XPIPE=0 NO_DMA+NO_HSTORE+D:XHORDER=0 NO_DMA+NO_HSTORE+D:XHORDER=0+D:XPIPE=0 NO_DMA+NO_HSTORE+NO_CELL+NO_REC NO_DMA+D:XSTAMP=3+D:XSTAMP_W=4+D:XHORDER=0 D:XHORDER=0 D:XHORDER=0+D:XPIPE=0 > gpurun_out/r06w/abl_clusterx_nodma.log 2>&1
echo rc=$?; cat gpurun_out/r06w/abl_clusterx_nodma.log
