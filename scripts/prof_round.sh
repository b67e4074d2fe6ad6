#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repo root):  scripts/prof_round.sh r04
#   headline: scripts/prof_zmp.sh;  secondary kernels: scripts/prof_kernel.sh with HBM=1 (kernel trace + stats, two SQ
#   counter passes, FETCH_SIZE and WRITE_SIZE passes -- every --pmc pass in its own run, never with a trace domain).
# scripts/summarize_round.py then writes the tracked summaries into profiles/.
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
bash scripts/prof_zmp.sh $TAG > gpurun_out/${TAG}_prof_zmp.log 2>&1
export HBM=1
bash scripts/prof_kernel.sh $TAG ddp python scripts/ddp_bench.py 4096 2 cen
bash scripts/prof_kernel.sh $TAG srb python scripts/ddp_bench.py 32768 2 srb
bash scripts/prof_kernel.sh $TAG walk python bench.py --workload walk --no-cpu-baseline --no-history-leg --no-live-counters --steps 2 --warmup 1
bash scripts/prof_kernel.sh $TAG multi python bench.py --workload multi --no-cpu-baseline --no-history-leg --no-live-counters --steps 2 --warmup 1
bash scripts/prof_kernel.sh $TAG xy python bench.py --workload xy --no-cpu-baseline --no-history-leg --no-live-counters --steps 5 --warmup 1
bash scripts/prof_zmp100.sh $TAG > gpurun_out/${TAG}_prof_zmp100.log 2>&1   # -> python scripts/summarize_zmp100.py $TAG (after summarize_round.py)
