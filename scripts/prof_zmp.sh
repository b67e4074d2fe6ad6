#!/bin/bash
# Profile the headline bench on the GPU box (run through gpurun from the repo root):
#   pass 1: rocprofv3 --kernel-trace --stats   (per-kernel durations)
#   pass 2..4: PMC counters, each in its own run (no trace domains combined with --pmc)
# Raw output goes to gpurun_out/<tag>_*/ ; scripts/summarize_prof.py turns it into profiles/<tag>_*.
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
# --no-host-p50: the p50 loops launch the same kernel on pinned HOST memory (PCIe-bound, 1.3 ms); left in, they are
# averaged into the kernel's --stats line, which is meant to be compared with bench.py's device-resident kernel_ms
B="python bench.py --no-cpu-baseline --no-host-p50 --no-secondary --no-history-leg --no-live-counters"
for d in trace pmc_sq1 pmc_sq2 pmc_fetch pmc_write; do mkdir -p gpurun_out/${TAG}_$d; done
# what the profiled library was built from (bench lines replay counters only for the same kernel sources)
python -m centroidalcontrolcollection_amd.build --kernel-hashes 2>/dev/null | tail -1 > gpurun_out/${TAG}_kernel_hashes.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -o zmp -- $B --steps 50 --warmup 5 > gpurun_out/${TAG}_trace/bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/${TAG}_pmc_sq1 -o zmp -- $B --steps 3 --warmup 1 > gpurun_out/${TAG}_pmc_sq1/bench.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${TAG}_pmc_sq2 -o zmp -- $B --steps 3 --warmup 1 > gpurun_out/${TAG}_pmc_sq2/bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_fetch -o zmp -- $B --steps 3 --warmup 1 > gpurun_out/${TAG}_pmc_fetch/bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_write -o zmp -- $B --steps 3 --warmup 1 > gpurun_out/${TAG}_pmc_write/bench.log 2>&1
find gpurun_out -name "*.csv" | head -30
