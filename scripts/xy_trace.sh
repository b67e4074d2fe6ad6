#!/bin/bash
# kernel trace (start / end of every dispatch) of one LinearMpcXY step: scripts/xy_trace.sh <tag> [ENV=..]   (GPU box)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
k=$1; shift
d=gpurun_out/xytrace_$k
mkdir -p $d
env "$@" rocprofv3 --kernel-trace --output-format csv -d $d -o k -- python scripts/xy_bench.py 65536 2 > $d/run.log 2>&1
python - "$d" <<'PY'
import sys, csv, glob
f = glob.glob(sys.argv[1] + "/**/k_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "xy_" in r["Kernel_Name"] or "order_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: from the last launch of the first kernel a call with a history starts with (or a third of the rows)
marks = [i for i, r in enumerate(rows) if "order_count" in r["Kernel_Name"]]
last = rows[marks[-1]:] if marks else rows[-(len(rows) // 3):]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-24s start %8.3f  end %8.3f  dur %7.3f ms  grid %s queue %s" % (r["Kernel_Name"].replace("ccc_amd::", "").replace("void ", "")[:24], (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X", ""), r.get("Queue_Id", "")))
print("step span %.3f ms" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e6))
PY
