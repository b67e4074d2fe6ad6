#!/usr/bin/env python
"""Turn the raw rocprofv3 output of scripts/prof_kernel.sh (gpurun_out/<tag>_<name>_*) into the tracked summaries
profiles/<tag>_<name>_kernel_stats.csv and profiles/<tag>_<name>_counters.csv.  usage: summarize_kernel.py tag name match"""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, match = sys.argv[1], sys.argv[2], sys.argv[3]
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "%s_%s_trace" % (tag, name), "**", "k_kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
rows = []
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    acc, meta = collections.defaultdict(list), {}
    found = glob.glob(os.path.join(src, "%s_%s_%s" % (tag, name, d), "**", "k_counter_collection.csv"), recursive=True)
    if not found:
        continue
    path = found[0]
    for r in csv.DictReader(open(path)):
        if match not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = dict(kernel=r["Kernel_Name"].split("(")[0], grid=r["Grid_Size"], wg=r["Workgroup_Size"],
                    lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"])
    for k, v in sorted(acc.items()):
        rows.append(dict(counter=k, avg_per_dispatch=sum(v) / len(v), dispatches=len(v), **meta))
with open(os.path.join(dst, "%s_%s_counters.csv" % (tag, name)), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
for r in rows:
    print("%-24s %.4g" % (r["counter"], r["avg_per_dispatch"]))
byc = {r["counter"]: r["avg_per_dispatch"] for r in rows}
if "FETCH_SIZE" in byc and "WRITE_SIZE" in byc:
    # MI355X_MICROARCH.md: both counters are in KiB; gfx950 counts a 128-byte fetch as 64 bytes -> FETCH_SIZE x 2
    print("HBM bytes per launch: fetch %.4g (x2 corrected) + write %.4g = %.4g" % (
        byc["FETCH_SIZE"] * 2048, byc["WRITE_SIZE"] * 1024, byc["FETCH_SIZE"] * 2048 + byc["WRITE_SIZE"] * 1024))
print(open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name))).read())
