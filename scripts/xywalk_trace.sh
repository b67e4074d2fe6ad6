#!/bin/bash
# dispatch timeline of one step of the 32-ridge LinearMpcXY workload (GPU box): three block rounds, then the single-change
# safeguard round (measured, round 4: 5.5 + 3.8 + 4.3 ms, then 21.3 ms for the few per cent of instances that wander)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
d=gpurun_out/xywalk_trace; mkdir -p $d
rocprofv3 --kernel-trace --output-format csv -d $d -o k -- python bench.py --workload xywalk --no-cpu-baseline --steps 2 --warmup 1 > $d/run.log 2>&1
python - "$d" <<'PY'
import sys, csv, glob
f = glob.glob(sys.argv[1] + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "xy_plan" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
last = rows[-n:]; t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-40s start %8.3f end %8.3f dur %7.3f grid %s" % (r["Kernel_Name"].replace("ccc_amd::","").replace("void ","")[:40], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Grid_Size_X","")))
PY
