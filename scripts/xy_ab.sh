#!/bin/bash
# A/B timing of library variants built by scripts/unit_variant.sh (on the GPU box):  scripts/xy_ab.sh base rb6 "lanes:CCC_XY_LANES=32" ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  k=${spec%%:*}; envs=""; [ "$spec" != "$k" ] && envs=${spec#*:}
  for n in ${XY_AB_N:-65536 8192}; do
    echo "== $k $envs n=$n"; env $envs CCC_AMD_LIB=$PWD/scratch/libccc_$k.so python scripts/xy_bench.py $n 5
  done
done
