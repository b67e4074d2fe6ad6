#!/bin/bash
# Development aid: a variant of the library that differs in the flags csrc/ddp_tile.hip is compiled with.
#   scripts/build_tile_variant.sh <name> [-D...]   ->  scratch/libccc_<name>.so   (run with CCC_AMD_LIB=...)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OBJ=$ROOT/scratch/obj_base
mkdir -p $OBJ
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function"
for f in $ROOT/centroidalcontrolcollection_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  [ "$b" = "ddp_tile" ] && continue
  if [ ! -f $OBJ/$b.o ] || [ $f -nt $OBJ/$b.o ]; then $HIPCC $FLAGS -c $f -o $OBJ/$b.o & fi
done
wait
$HIPCC $FLAGS "$@" -c $ROOT/centroidalcontrolcollection_amd/csrc/ddp_tile.hip -o $ROOT/scratch/ddp_tile_$NAME.o
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJ/*.o $ROOT/scratch/ddp_tile_$NAME.o -o $ROOT/scratch/libccc_$NAME.so
echo built scratch/libccc_$NAME.so
