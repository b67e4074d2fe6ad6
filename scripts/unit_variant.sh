#!/bin/bash
# Build a variant of ONE translation unit (extra -D switches) and link it with the stock objects of the others into
# scratch/libccc_<name>.so, for A/B timing on the GPU box (CCC_AMD_LIB=scratch/libccc_<name>.so python scripts/...).
# usage: scripts/unit_variant.sh <unit> <name> [-Dflags ...]      e.g.  scripts/unit_variant.sh xy base
set -e
cd "$(dirname "$0")/.."
UNIT=$1; NAME=$2; shift 2
CS=centroidalcontrolcollection_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
mkdir -p scratch/obj/stock scratch/obj/var
for f in $CS/*.hip; do
  b=$(basename $f .hip)
  [ "$b" = "$UNIT" ] && continue
  if [ ! -f scratch/obj/stock/$b.o ] || [ $f -nt scratch/obj/stock/$b.o ]; then /opt/rocm/bin/hipcc $FL -c $f -o scratch/obj/stock/$b.o & fi
done
/opt/rocm/bin/hipcc $FL "$@" -c $CS/$UNIT.hip -o scratch/obj/var/${UNIT}_$NAME.o
wait
OBJS=$(ls scratch/obj/stock/*.o | grep -v "/${UNIT}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS scratch/obj/var/${UNIT}_$NAME.o -o scratch/libccc_$NAME.so
echo built scratch/libccc_$NAME.so
