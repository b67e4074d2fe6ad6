cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/r01_trace gpurun_out/r01_pmc1 gpurun_out/r01_pmc2
rocprofv3 --kernel-trace --stats -d gpurun_out/r01_trace -o zmp -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r01_trace/bench.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/r01_pmc1 -o zmp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01_pmc1/bench.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d gpurun_out/r01_pmc2 -o zmp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01_pmc2/bench.log 2>&1
ls -R gpurun_out | head -40
