#!/bin/bash
# round 6, first GPU call: the new evidence tests (full-size step as one RCCL rank; C4 sampler waveform; C4 T = 501 backward) + start-of-round bench line
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_train_gpu.py -m gpu -x -q -k "full_size_c2_step_as_one_rccl_rank" > $O/test_rccl.log 2>&1; echo "rccl rc=$?"
python -m pytest tests/test_c4_fullsize_gpu.py -m gpu -q -s -k "(enhance and not f16) or zz" > $O/test_c4.log 2>&1; echo "c4 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/test_rccl.log; tail -5 $O/test_c4.log; cut -c1-600 $O/bench.json
