#!/bin/bash
# Development aid (GPU box): walking (32 ridges) and multi-contact (64 ridges) bench lines for library variants.
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export CCC_AMD_LIB=$PWD/scratch/libccc_$v.so
  echo "== $v"
  for w in walk multi; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$w', round(d['value']), d['ms_per_step'])"; done
done
