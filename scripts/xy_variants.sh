#!/bin/bash
# Build profiling variants of the library (extra -D switches for csrc/xy.hip) into scratch/ and time each on the GPU box.
# usage (here):   scripts/xy_variants.sh build
#       (on GPU): scripts/xy_variants.sh run [n]
set -e
cd "$(dirname "$0")/.."
SRC="centroidalcontrolcollection_amd/csrc/*.hip"
declare -A V=( [full]="" [setup_only]="-DXY_ROUNDS=0" [no_adds]="-DXY_ROUNDS=0 -DXY_PROF_NO_ADDS" )
if [ "$1" = build ]; then
  mkdir -p scratch
  for k in "${!V[@]}"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${V[$k]} $SRC -o scratch/libccc_xy_$k.so
  done
else
  for k in full setup_only no_adds; do
    echo "== $k"; CCC_AMD_LIB=$PWD/scratch/libccc_xy_$k.so python scripts/xy_bench.py ${2:-4096} 3
  done
fi
