#!/bin/bash
# A/B of the DDP scheduler's slice lengths on the GPU box (development switches CCC_DDP_SLICE="first,next", CCC_DDP_SLOTS)
cd $GRAFT_REPO_ROOT
echo "== correctness with forced slicing (32 slots, slices 2,1)"
CCC_DDP_SLOTS=32 CCC_DDP_SLICE=2,1 python -m pytest tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | head -5
for sl in ${DDP_AB_SLICES:-0 2,2 2,4 2,8 1,4 3,6 4,20}; do
  echo "== slice $sl"
  CCC_DDP_SLICE=$sl python scripts/ddp_bench.py 4096 3 cen 2>&1 | grep solves
  CCC_DDP_SLICE=$sl python scripts/ddp_bench.py 4096 3 srb 2>&1 | grep solves
  CCC_DDP_SLICE=$sl python scripts/ddp_bench.py 32768 3 srb 2>&1 | grep solves
done
