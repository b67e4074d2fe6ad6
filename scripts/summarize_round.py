#!/usr/bin/env python
"""Turn the raw rocprofv3 output of scripts/prof_round.sh (gpurun_out/<tag>_*) into the tracked summaries of a round:
profiles/<tag>_zmp_* (scripts/summarize_prof.py), profiles/<tag>_<name>_{kernel_stats,counters}.csv per secondary kernel
(scripts/summarize_kernel.py), and the two files the bench lines read:
  profiles/<tag>_ddp_valu_counters.json   VALU issue share / wait share / instructions per instance of the DDP kernels
  profiles/<tag>_hbm_traffic.json         measured HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) per secondary workload
usage: summarize_round.py r04"""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
py = sys.executable
subprocess.check_call([py, os.path.join(ROOT, "scripts", "summarize_prof.py"), tag], stdout=subprocess.DEVNULL)
MATCH = dict(ddp="ddp_tile_kernel<9, 1>", srb="ddp_tile_kernel<12, 1>", walk="ddp_tile_kernel<9, 2>", multi="ddp_tile_kernel<9, 4>")
KEY = dict(ddp="S9", srb="S12", walk="S9M32", multi="S9M64")
BATCH = dict(ddp=4096, srb=32768, walk=4096, multi=2048)


def counters(name):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_%s_counters.csv" % (tag, name)))))
    return {r["counter"]: float(r["avg_per_dispatch"]) for r in rows}, rows[0]


def kernel_ms(name, match):
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, name)))):
        if match in r["Name"]:
            return float(r["AverageNs"]) * 1e-6
    raise SystemExit("no %s in the kernel stats of %s" % (match, name))


# the kernel sources the profiled library was built from (written on the GPU box by scripts/prof_kernel.sh): the bench
# lines replay these counters only when their own library was built from the same sources
with open(os.path.join(ROOT, "gpurun_out", tag + "_kernel_hashes.json")) as f:
    KH = json.load(f)
valu, traffic = {"kernel_hash": KH["ddp"]}, {}
for name, match in MATCH.items():
    subprocess.check_call([py, os.path.join(ROOT, "scripts", "summarize_kernel.py"), tag, name, match], stdout=subprocess.DEVNULL)
    c, meta = counters(name)
    ms = kernel_ms(name, match)
    valu[KEY[name]] = dict(valu_issue_frac=c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * ms * 1e6 * 2.4),
                           wait_any_frac=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                           # share of a wavefront's resident time in which it executes a VALU instruction, and -- x the
                           # wavefronts that share a SIMD (persistent grid: SQ_WAVES / 1024 SIMDs) -- how busy the SIMD's
                           # one VALU is.  Independent of the clock (the 2.4 GHz in valu_issue_frac is the PEAK clock; these
                           # kernels run at ~1.3 GHz: SQ_WAVE_CYCLES x 4 / waves / kernel time)
                           valu_active_frac_of_wave=c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                           waves_per_simd=c["SQ_WAVES"] / 1024.0,
                           simd_valu_busy_frac=c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * c["SQ_WAVES"] / 1024.0,
                           effective_clock_ghz=c["SQ_WAVE_CYCLES"] * 4.0 / c["SQ_WAVES"] / (ms * 1e6),
                           valu_insts_per_instance=c["SQ_INSTS_VALU"] / BATCH[name],
                           salu_insts_per_instance=c["SQ_INSTS_SALU"] / BATCH[name],
                           lds_insts_per_instance=c["SQ_INSTS_LDS"] / BATCH[name],
                           # (rocprofv3's VGPR_Count is the ARCHITECTURAL vector registers; the kernel also holds accumulation
                           #  registers -- the compiler's total is pinned in tests/test_kernel_resources.py.  VERDICT r4 item 8)
                           kernel_ms=ms, arch_vgpr=meta["vgpr"], scratch=meta["scratch"], lds=meta["lds"],
                           lds_bank_conflict_frac=c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
                           source="profiles/%s_%s_counters.csv + %s_%s_kernel_stats.csv (batch %d)" % (tag, name, tag, name, BATCH[name]))
    if "FETCH_SIZE" in c:
        traffic[name] = dict(batch=BATCH[name], kernel=match, kernel_hash=KH[name], fetch_bytes_corrected=c["FETCH_SIZE"] * 2048,
                             write_bytes=c["WRITE_SIZE"] * 1024, hbm_bytes_per_launch=c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024)
json.dump(valu, open(os.path.join(ROOT, "profiles", "%s_ddp_valu_counters.json" % tag), "w"), indent=1)
# LinearMpcXY: a step is several dispatches (three rounds of the stage kernel + the dual kernel on the hand-over lists):
# per-dispatch averages x dispatches per step, per kernel
for kern, short in (("xy_plan_stream_kernel", "xystream"), ("xy_plan_kernel", "xydual")):
    subprocess.check_call([py, os.path.join(ROOT, "scripts", "summarize_kernel.py"), tag, "xy", kern], stdout=subprocess.DEVNULL)
    for ext in ("counters", ):
        os.replace(os.path.join(ROOT, "profiles", "%s_xy_%s.csv" % (tag, ext)), os.path.join(ROOT, "profiles", "%s_%s_%s.csv" % (tag, short, ext)))
steps_profiled = 3 + 1 + 1  # timed + warm-up steps of the profiled command + the untimed status launch of bench_secondary
tot = 0.0
xy = {}
for short in ("xystream", "xydual"):
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_%s_counters.csv" % (tag, short)))))
    c = {r["counter"]: (float(r["avg_per_dispatch"]), int(r["dispatches"])) for r in rows}
    if "FETCH_SIZE" in c:
        per_step = (c["FETCH_SIZE"][0] * 2048 * c["FETCH_SIZE"][1] + c["WRITE_SIZE"][0] * 1024 * c["WRITE_SIZE"][1])
        xy[short] = dict(fetch_bytes_corrected_per_dispatch=c["FETCH_SIZE"][0] * 2048, write_bytes_per_dispatch=c["WRITE_SIZE"][0] * 1024,
                         dispatches_in_the_profiled_run=c["FETCH_SIZE"][1])
        tot += per_step
if xy:
    # dispatches per step: count from the kernel stats (calls of the profiled run / launches of a step)
    traffic["xy"] = dict(batch=65536, kernel_hash=KH["xy"], kernels=xy, total_bytes_in_the_profiled_run=tot,
                         note="divide by the batched calls of the profiled run (bench.py --workload xy --steps 5 --warmup 1: the first call without an order: "
                              "see <tag>_xy_kernel_stats.csv for the calls per kernel) for bytes per 65536-instance step")
    calls = None
    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_xy_kernel_stats.csv" % tag))):
        if "xy_plan_stream_kernel" in r["Name"]:
            calls = int(r["Calls"])
    if calls:
        traffic["xy"]["stream_kernel_calls"] = calls
        traffic["xy"]["steps_in_the_profiled_run"] = calls / 3.0  # three rounds per step
        traffic["xy"]["hbm_bytes_per_step"] = tot / (calls / 3.0)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % tag), "w"), indent=1)
print(json.dumps(valu, indent=1))
print(json.dumps(traffic, indent=1))
