#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zc; mkdir -p $O
URSE_LIB_PATH=variants/liburse_pr2d6.so timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_c2_parity_gpu.py -x -q -m gpu -k "rounds or fused_projection" > $O/test_cx.log 2>&1
echo "cx tests (pr2d6) rc=$?"; tail -3 $O/test_cx.log
timeout 1200 python scripts/abl_clusterx.py D:XSTAMP=3+D:XPROJ_RING=0 D:XSTAMP=3 D:XSTAMP=3+D:XPROJ_RING=2+D:XPD=6 D:XSTAMP=3+D:XPROJ_RING=2+D:XPD=4 D:XSTAMP=3+D:XPROJ_RING=2+D:XPD=6+D:XSTAMP_W=4 D:XSTAMP=3+D:XPROJ_RING=0+D:XSTAMP_W=4 > $O/abl_clusterx_proj.log 2>&1
echo rc=$?; grep -v "^   \(gathered\|at b1\|after b2\|published\)" $O/abl_clusterx_proj.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_pr0.so" "URSE_LIB_PATH=variants/liburse_pr2d6.so" "URSE_LIB_PATH=variants/liburse_pr2d4.so" > $O/ab_proj_ring.log 2>&1
cat $O/ab_proj_ring.log
