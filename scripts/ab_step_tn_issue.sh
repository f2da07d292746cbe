#!/bin/bash
# same-box A/B of the train step: library as built (lean TN issue path) vs a build with -DURSE_TN_LEAN_ISSUE=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
CS=$R/urgent2026_challenge_track1_amd/csrc
mkdir -p /tmp/altlib
for f in $CS/*.hip; do
  n=$(basename $f .hip)
  fl=""
  [ "$n" = "gemm" ] && fl="-DURSE_TN_LEAN_ISSUE=0"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value $fl -c $f -o /tmp/altlib/$n.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/altlib/liburse_generic.so /tmp/altlib/*.o
B="python $R/bench.py --no-flow --no-f32-mode --no-dist-leg --no-metrics --no-cpu-baseline --steps 8 --warmup 3"
for i in 1 2 3; do
  echo "lean:    $($B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  echo "generic: $(URSE_LIB_PATH=/tmp/altlib/liburse_generic.so $B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done
