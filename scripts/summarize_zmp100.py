#!/usr/bin/env python
"""Turn the raw rocprofv3 output of scripts/prof_zmp100.sh (gpurun_out/<tag>_zmp100_*) into profiles/<tag>_zmp100_kernel_stats.csv,
profiles/<tag>_zmp100_counters.csv (the state-space kernel) and profiles/<tag>_zmp100list_counters.csv (the exact kernel on
its hand-over list), and add the measured HBM bytes of one step -- both kernels -- to profiles/<tag>_hbm_traffic.json as
"zmp100" (what the bench line of that workload replays as roofline.traffic).  usage: summarize_zmp100.py r06"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
py, prof = sys.executable, os.path.join(ROOT, "profiles")
tot, per = 0.0, {}
for match, short in (("zmp_plan_reg_list_kernel", "zmp100list"), ("zmp_plan_stage_kernel", "zmp100")):
    subprocess.check_call([py, os.path.join(ROOT, "scripts", "summarize_kernel.py"), tag, "zmp100", match], stdout=subprocess.DEVNULL)
    if short != "zmp100":
        os.replace(os.path.join(prof, "%s_zmp100_counters.csv" % tag), os.path.join(prof, "%s_%s_counters.csv" % (tag, short)))
    c = {r["counter"]: float(r["avg_per_dispatch"]) for r in csv.DictReader(open(os.path.join(prof, "%s_%s_counters.csv" % (tag, short))))}
    per[short] = dict(kernel=match, fetch_bytes_corrected=c["FETCH_SIZE"] * 2048, write_bytes=c["WRITE_SIZE"] * 1024)
    tot += c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024
with open(os.path.join(ROOT, "gpurun_out", tag + "_kernel_hashes.json")) as f:
    kh = json.load(f)["zmp100"]
path = os.path.join(prof, "%s_hbm_traffic.json" % tag)
traffic = json.load(open(path))
traffic["zmp100"] = dict(batch=32768, kernel="zmp_plan_stage_kernel + zmp_plan_reg_list_kernel", kernel_hash=kh, kernels=per,
                         hbm_bytes_per_step=tot,
                         note="one step = one launch of each kernel (bench.py --workload zmp100); FETCH_SIZE x 2 per "
                              "MI355X_MICROARCH.md, separate --pmc passes")
json.dump(traffic, open(path, "w"), indent=1)
print(json.dumps(traffic["zmp100"], indent=1))
