#!/bin/bash
# Development aid (GPU box): time the variants built by scripts/build_tile_variant.sh on the DDP bench shapes.
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export CCC_AMD_LIB=$PWD/scratch/libccc_$v.so
  echo "== $v"
  python scripts/ddp_bench.py 4096 3 cen | tail -1
  python scripts/ddp_bench.py 1024 3 cen | tail -1
  python scripts/ddp_bench.py 32768 2 srb | tail -1
  python scripts/ddp_bench.py 4096 3 srb | tail -1
done
