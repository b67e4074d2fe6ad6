cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
python scripts/ddp_bench.py 4096 2
mkdir -p gpurun_out/ddp_pmc
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/ddp_pmc -o d1 -- python scripts/ddp_bench.py 1024 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT --output-format csv -d gpurun_out/ddp_pmc -o d2 -- python scripts/ddp_bench.py 1024 1 > /dev/null 2>&1
python - <<'PY'
import csv, collections
for f in ('gpurun_out/ddp_pmc/d1_counter_collection.csv','gpurun_out/ddp_pmc/d2_counter_collection.csv'):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ddp_plan' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
