#!/bin/bash
# kernel trace (start / end of every dispatch) of one LinearMpcXY step for a library variant: scripts/xy_trace.sh <variant> [ENV=..]
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
k=$1; shift
d=gpurun_out/xytrace_$k$(echo "$*" | tr -c 'A-Za-z0-9\n' '_')
mkdir -p $d
env "$@" CCC_AMD_LIB=$PWD/scratch/libccc_$k.so rocprofv3 --kernel-trace --output-format csv -d $d -o k -- python scripts/xy_bench.py 65536 2 > $d/run.log 2>&1
python - "$d" <<'PY'
import sys, csv, glob
f = glob.glob(sys.argv[1] + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "xy_plan" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-5:]   # the last step's five dispatches
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-28s start %8.3f ms  dur %7.3f ms  grid %s" % (r["Kernel_Name"][:28].replace("ccc_amd::", ""), (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "")))
print("step span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
PY
