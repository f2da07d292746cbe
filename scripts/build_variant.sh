#!/bin/bash
# build variants/liburse_<name>.so with extra flags on ONE source file (the other objects come from the in-tree build):
#   bash scripts/build_variant.sh kb12 lstm "-DURSE_BWD_KB=12"   ->  URSE_LIB_PATH=variants/liburse_kb12.so python bench.py ...
name=$1; file=$2; flags=$3
R=$(cd $(dirname $0)/.. && pwd)
P=$R/urgent2026_challenge_track1_amd
mkdir -p $R/variants/$name
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value $flags -c $P/csrc/$file.hip -o $R/variants/$name/$file.o || exit 1
objs=""
for o in $P/build/*.o; do
  n=$(basename $o)
  if [ "$n" = "$file.o" ]; then objs="$objs $R/variants/$name/$file.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/liburse_$name.so $objs && echo built $R/variants/liburse_$name.so
