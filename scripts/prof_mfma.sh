#!/bin/bash
# MFMA counters of one kernel (its own PMC pass): scripts/prof_mfma.sh <tag> <name> <command...>
TAG=$1; NAME=$2; shift 2
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
mkdir -p gpurun_out/${TAG}_${NAME}_mfma
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${TAG}_${NAME}_mfma -o k -- "$@" > gpurun_out/${TAG}_${NAME}_mfma/run.log 2>&1
tail -2 gpurun_out/${TAG}_${NAME}_mfma/run.log
