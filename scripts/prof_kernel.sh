#!/bin/bash
# Profile one of the secondary kernels on the GPU box (run through gpurun from the repo root):
#   scripts/prof_kernel.sh <tag> <name> <command...>      e.g.  scripts/prof_kernel.sh r01 xy python scripts/xy_bench.py 16384 3
# pass 1: rocprofv3 --kernel-trace --stats; pass 2-3: PMC counters, each in its own run (never combined with a trace
# domain).  Raw output: gpurun_out/<tag>_<name>_*/ ; scripts/summarize_kernel.py copies the summaries to profiles/.
TAG=$1; NAME=$2; shift 2
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
for d in trace pmc1 pmc2; do mkdir -p gpurun_out/${TAG}_${NAME}_$d; done
python -m centroidalcontrolcollection_amd.build --kernel-hashes 2>/dev/null | tail -1 > gpurun_out/${TAG}_kernel_hashes.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_${NAME}_trace -o k -- "$@" > gpurun_out/${TAG}_${NAME}_trace/run.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/${TAG}_${NAME}_pmc1 -o k -- "$@" > gpurun_out/${TAG}_${NAME}_pmc1/run.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/${TAG}_${NAME}_pmc2 -o k -- "$@" > gpurun_out/${TAG}_${NAME}_pmc2/run.log 2>&1
# HBM=1: two more passes for the HBM traffic (FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction in summarize_kernel.py)
if [ -n "$HBM" ]; then
  for d in pmc3 pmc4; do mkdir -p gpurun_out/${TAG}_${NAME}_$d; done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_${NAME}_pmc3 -o k -- "$@" > gpurun_out/${TAG}_${NAME}_pmc3/run.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_${NAME}_pmc4 -o k -- "$@" > gpurun_out/${TAG}_${NAME}_pmc4/run.log 2>&1
fi
tail -2 gpurun_out/${TAG}_${NAME}_trace/run.log
