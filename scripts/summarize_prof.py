#!/usr/bin/env python
"""Turn the raw rocprofv3 output of scripts/prof_zmp.sh (gpurun_out/<tag>_*) into the tracked summaries
profiles/<tag>_zmp_kernel_stats.csv, profiles/<tag>_zmp_counters.csv and profiles/zmp_hbm_traffic.json
(the per-launch HBM byte count bench.py reports as roofline.traffic)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
# the kernel sources the profiled library was built from (written on the GPU box by scripts/prof_zmp.sh)
with open(os.path.join(src, tag + "_kernel_hashes.json")) as f:
    KH = json.load(f)["zmp"]

shutil.copy(os.path.join(src, tag + "_trace", "zmp_kernel_stats.csv"), os.path.join(dst, tag + "_zmp_kernel_stats.csv"))
rows = []
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(list)
    meta = {}
    path = os.path.join(src, "%s_%s" % (tag, d), "zmp_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        # the headline launches: the static-pairing kernel on a handle with a history (round 5; bench.py's leg without a
        # history runs zmp_plan_kernel_dyn, the ordering itself is zmp_order_kernel -- both have lines of their own in the
        # kernel statistics)
        if "zmp_plan_kernel<" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = dict(kernel=r["Kernel_Name"].split("(")[0], grid=r["Grid_Size"], wg=r["Workgroup_Size"],
                    lds=r["LDS_Block_Size"], vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], scratch=r["Scratch_Size"])
    for k, v in sorted(acc.items()):
        rows.append(dict(counter=k, avg_per_dispatch=sum(v) / len(v), dispatches=len(v), **meta))
with open(os.path.join(dst, tag + "_zmp_counters.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
c = {r["counter"]: r["avg_per_dispatch"] for r in rows}
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of the bytes of a
# coalesced streaming read (128-B requests tallied at 64 B) -> doubled.  WRITE_SIZE is uncalibrated and taken as is.
fetch = c["FETCH_SIZE"] * 1024 * 2
write = c["WRITE_SIZE"] * 1024
out = dict(tag=tag, kernel_hash=KH, workload="LinearMpcZmp N=32 batch=65536", kernel=rows[0]["kernel"],
           fetch_size_kib_raw=c["FETCH_SIZE"], write_size_kib_raw=c["WRITE_SIZE"],
           fetch_bytes_corrected=fetch, write_bytes=write, hbm_bytes_per_launch=fetch + write,
           algorithmic_bytes_per_launch=1088 * 65536,
           note="FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); separate --pmc passes")
json.dump(out, open(os.path.join(dst, "zmp_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
# VALU-issue share (bench.py roofline.valu): a wave64 VALU instruction occupies its SIMD's 16-lane pipe for 4 clocks
# (fp64 FMA at full rate); 256 CUs x 4 SIMDs; clock from the kernel trace's average duration at 2.4 GHz
dur_ns = None
for r in csv.DictReader(open(os.path.join(dst, tag + "_zmp_kernel_stats.csv"))):
    if "zmp_plan_kernel<" in r["Name"]:
        dur_ns = float(r["AverageNs"])
        break
valu = dict(tag=tag, kernel_hash=KH, batch=65536, kernel=rows[0]["kernel"], kernel_avg_ns=dur_ns, sq_insts_valu=c["SQ_INSTS_VALU"],
            sq_insts_salu=c.get("SQ_INSTS_SALU"), sq_insts_lds=c.get("SQ_INSTS_LDS"),
            valu_insts_per_solve=c["SQ_INSTS_VALU"] / 65536.0,
            valu_issue_frac=c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * dur_ns * 2.4),
            note="SQ_INSTS_VALU (wave-instructions per launch) x 4 clk / (1024 SIMDs x kernel clocks at 2.4 GHz)")
json.dump(valu, open(os.path.join(dst, "zmp_valu_counters.json"), "w"), indent=1)
print(json.dumps(valu, indent=1))
for r in rows:
    print("%-24s %16.1f" % (r["counter"], r["avg_per_dispatch"]))
