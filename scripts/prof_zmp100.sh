#!/bin/bash
# LinearMpcZmp at N = 100 (the state-space kernel KS + the exact kernel on its hand-over list): kernel trace + stats, two SQ
# counter passes, FETCH_SIZE / WRITE_SIZE passes (scripts/prof_kernel.sh).  usage (through gpurun): scripts/prof_zmp100.sh r06
TAG=${1:-r06}
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
HBM=1 bash scripts/prof_kernel.sh $TAG zmp100 python bench.py --workload zmp100 --no-cpu-baseline --no-history-leg --no-live-counters --steps 10 --warmup 2
