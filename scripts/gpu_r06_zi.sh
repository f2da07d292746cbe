#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zi; mkdir -p $O
URSE_LIB_PATH=variants/liburse_pe1.so timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_c2_parity_gpu.py -x -q -m gpu -k "rounds or fused_projection" > $O/test_cx.log 2>&1; echo "tests (pe1) rc=$?"; tail -2 $O/test_cx.log
timeout 1200 python scripts/abl_clusterx.py D:XSTAMP=3 D:XSTAMP=3+D:XPUB_EARLY=1 D:XSTAMP=3+D:XPUB_EARLY=1+D:XSTAMP_W=4 D:XSTAMP=3+D:XSTAMP_W=4 > $O/abl_clusterx_pub.log 2>&1
echo rc=$?; grep -v "^   \(gathered\|at b1\|issued\)" $O/abl_clusterx_pub.log
timeout 2400 bash scripts/ab_step_sets.sh "-" "URSE_LIB_PATH=variants/liburse_pe1.so" "URSE_LIB_PATH=variants/liburse_pe1s24.so" "URSE_LIB_PATH=variants/liburse_pe1s0.so" > $O/ab_pub_early.log 2>&1
cat $O/ab_pub_early.log
