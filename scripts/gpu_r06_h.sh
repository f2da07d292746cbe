#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
python scripts/stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
