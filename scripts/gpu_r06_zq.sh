#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r06zq; mkdir -p $O
timeout 1200 python -m pytest tests/test_entry_gpu.py tests/test_host_cpu.py -x -q -m gpu > $O/test_bench.log 2>&1; echo "rc=$?"; tail -3 $O/test_bench.log
