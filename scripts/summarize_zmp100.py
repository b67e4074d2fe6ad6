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
# what the state-space kernel is bound by: its own vector instructions on one wavefront per SIMD
c = {r["counter"]: float(r["avg_per_dispatch"]) for r in csv.DictReader(open(os.path.join(prof, "%s_zmp100_counters.csv" % tag)))}
ms = None
for r in csv.DictReader(open(os.path.join(prof, "%s_zmp100_kernel_stats.csv" % tag))):
    if "zmp_plan_stage_kernel" in r["Name"]:
        ms = float(r["AverageNs"]) * 1e-6
valu = dict(kernel_hash=kh, batch=32768, kernel="zmp_plan_stage_kernel", kernel_ms=ms,
            simd_valu_busy_frac=c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * c["SQ_WAVES"] / 1024.0,
            wait_frac=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], waves_per_simd=c["SQ_WAVES"] / 1024.0,
            valu_insts_per_wavefront=c["SQ_INSTS_VALU"] / c["SQ_WAVES"], salu_insts_per_wavefront=c["SQ_INSTS_SALU"] / c["SQ_WAVES"],
            effective_clock_ghz=c["SQ_WAVE_CYCLES"] * 4.0 / c["SQ_WAVES"] / (ms * 1e6),
            source="profiles/%s_zmp100_counters.csv + %s_zmp100_kernel_stats.csv (batch 32768)" % (tag, tag))
json.dump(valu, open(os.path.join(prof, "%s_zmp100_valu_counters.json" % tag), "w"), indent=1)
print(json.dumps(valu, indent=1))
