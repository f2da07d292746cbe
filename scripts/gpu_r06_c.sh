#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
( python scripts/exp_stft_cold.py
  for pp in 1 4; do EXP_YARDSTICKS=0 URSE_STFT960_PP=$pp python scripts/exp_stft_cold.py; done
  EXP_YARDSTICKS=0 URSE_STFT960_PIPE=0 python scripts/exp_stft_cold.py
  for v in nostore noload nodft neither lsonly; do EXP_YARDSTICKS=0 python scripts/exp_stft_cold.py variants/liburse_st_$v.so; done
  for v in nostore noload neither lsonly; do EXP_YARDSTICKS=0 URSE_STFT960_PP=4 python scripts/exp_stft_cold.py variants/liburse_st_$v.so; done
) 2>&1 | tee $O/exp_stft_cold.log
# gnb dgrad without its 6 spilled registers: this library vs the round-5 library under the same python, both orders
bash scripts/ab_step_sets.sh "URSE_LIB_PATH=variants/r05tree/urgent2026_challenge_track1_amd/liburse_hip.so" "-" 2>&1 | tee $O/ab_lib_r05_vs_now.log
