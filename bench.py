#!/usr/bin/env python
"""bench.py -- headline benchmark of the batched LinearMpcZmp planOnce() path on MI355X.

Metric (BASELINE.json): planOnce() solves/sec whole-node + p50 latency, LinearMpcZmp N=32 batch=65536, fp64.
A "step" is one pass of the hot path over one batch of synthetic instances already resident in HBM:
one launch of the dual active-set kernel (csrc/zmp.hip) through the C-ABI (ccc_zmp_plan_batch_device),
plus -- for N > 1 GPUs -- the RCCL all-gather of the planned ZMPs (north_star).  Weak scaling: every rank
solves its own batch of 65536 instances (seed = 20250928 + rank); value = all ranks' solves / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline] [--workload zmp|zmp100|xy|ddp|srb|ism|z|ddpzmp]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ALGO_BYTES_PER_SOLVE = 1088  # 48 B state + 1024 B limits in, 16 B ZMP out
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: vector fp64 (256 CUs x 64 lanes x 2 flop x 2.4 GHz)
# algorithmic fp64 work of one pivot of one QP: the symmetric rank-1 update of the 32 x 32 sweep
# tableau = 32 x 32 FMAs = 2048 flop; the ratio test, reciprocal and selection are O(32) and not counted
FLOP_PER_PIVOT = 2048


def _same_build(d):
    """The profile summary d was made from the kernel sources this library was built from (kernel_hash of
    centroidalcontrolcollection_amd/build.py: the csrc files of the headline kernel + the compiler flags)?  Counters of
    another build are not replayed."""
    from centroidalcontrolcollection_amd import build as _b

    return d.get("kernel_hash") == _b.kernel_hash("zmp")


def _same_kernel(d, kernel):
    """The summary is of the kernel the timed launches ran (names as the library / rocprofv3 spell them)."""
    return kernel is None or kernel.replace(" ", "") in d.get("kernel", "").replace(" ", "")


def measured_traffic(n, kernel=None):
    """(HBM bytes per launch | None, source) from the PMC passes of scripts/prof_zmp.sh (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE,
    summarised by scripts/summarize_prof.py into profiles/zmp_hbm_traffic.json); None if not collected for this batch or
    collected on another build of the kernel."""
    from centroidalcontrolcollection_amd import build as _b

    path = os.path.join(ROOT, "profiles", "zmp_hbm_traffic.json")
    try:
        d = json.load(open(path))
        if d.get("algorithmic_bytes_per_launch") == ALGO_BYTES_PER_SOLVE * n:
            if not _same_kernel(d, kernel):
                return None, "profiles/zmp_hbm_traffic.json refused: it profiled %s, the timed launches ran %s" % (d.get("kernel"), kernel)
            if _same_build(d):
                return d["hbm_bytes_per_launch"], "profiles/zmp_hbm_traffic.json (replayed; kernel_hash %s = this build)" % d["kernel_hash"]
            return None, "profiles/zmp_hbm_traffic.json refused: profiled build %s, this build %s" % (
                d.get("kernel_hash"), _b.kernel_hash("zmp"))
    except Exception:
        pass
    return None, None


def valu_counters(n, kernel=None):
    """VALU-issue share of the kernel from the PMC pass of the same launch (profiles/zmp_valu_counters.json, written by
    scripts/summarize_prof.py from rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES ...); None if not collected for this batch
    or collected on another build of the kernel."""
    path = os.path.join(ROOT, "profiles", "zmp_valu_counters.json")
    try:
        d = json.load(open(path))
        if d.get("batch") == n and _same_build(d) and _same_kernel(d, kernel):
            return d
    except Exception:
        pass
    return None


def parse_counter_csv(path, kernel):
    """rocprofv3 `*_counter_collection.csv` -> ({counter: average value per dispatch}, average dispatch duration in ns,
    dispatches) over the dispatches of `kernel` (names compared without spaces, as the library / rocprofv3 spell them)."""
    import collections
    import csv

    want = kernel.replace(" ", "")
    acc, dur = collections.defaultdict(dict), {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if want not in r["Kernel_Name"].replace(" ", ""):
                continue
            d = r["Dispatch_Id"]
            acc[r["Counter_Name"]][d] = acc[r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
            dur[d] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not dur:
        return {}, None, 0
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}, sum(dur.values()) / len(dur), len(dur)


def pmc_passes(inner, passes, env=None):
    """Run `inner` (a command list) once per entry of `passes` (lists of counter names) under `rocprofv3 --pmc ...`, each
    pass its own run with no trace domain beside it.  Returns ({counter: {kernel name: [sum over its dispatches, dispatches,
    sum of their durations in ns]}}, None) or (None, reason)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = {}
    for pmc in passes:
        d = tempfile.mkdtemp(prefix="ccc_live_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc"] + pmc + ["--output-format", "csv", "-d", d, "-o", "live", "--"] + inner, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp", **(env or {})), timeout=900, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(d, "**", "live_counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 wrote no counter file for %s" % pmc
            per = collections.defaultdict(lambda: collections.defaultdict(dict))
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    k = per[r["Counter_Name"]][r["Kernel_Name"].split("(")[0]]
                    e = k.setdefault(r["Dispatch_Id"], [0.0, float(r["End_Timestamp"]) - float(r["Start_Timestamp"])])
                    e[0] += float(r["Counter_Value"])
            for c, kernels in per.items():
                out[c] = {kn: [sum(v[0] for v in dd.values()), len(dd), sum(v[1] for v in dd.values())] for kn, dd in kernels.items()}
        except Exception as e:  # noqa: BLE001 -- a profiler that is missing or refuses must not fail the bench line
            return None, "live counter pass %s failed: %s" % (pmc[0], e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, None


def live_counters(batch, kernel):
    """the counters of the headline kernel measured IN THIS RUN.  A counter needs rocprofv3 around the
    process, so rank 0 runs this script again three times under `rocprofv3 --pmc ...` (each --pmc pass in its own run, no
    trace domain beside it: the guide's HBM recipe) as `--inner`: 2 rotating batches of the same size, ~130 launches of
    the same kernel (100 of them the clock ramp), nothing else.  Returns (dict | None, source string)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    inner = [sys.executable, os.path.abspath(__file__), "--inner", "--batch", str(batch), "--steps", "20", "--warmup", "5",
             "--ramp-steps", "100", "--ring", "2", "--no-cpu-baseline", "--no-host-p50", "--no-history-leg", "--no-secondary"]
    got, n_disp = {}, 0
    t_begin = time.perf_counter()
    for pmc in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"]):
        d = tempfile.mkdtemp(prefix="ccc_live_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc"] + pmc + ["--output-format", "csv", "-d", d, "-o", "live", "--"] + inner, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), timeout=600, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(d, "**", "live_counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 wrote no counter file for %s" % pmc
            c, dur, nd = parse_counter_csv(files[0], kernel)
            if not nd:
                return None, "no dispatch of %s in the %s pass" % (kernel, pmc[0])
            got.update(c)
            n_disp = nd
            if "SQ_INSTS_VALU" in c:
                got["dispatch_ns"] = dur
        except Exception as e:  # noqa: BLE001 -- a profiler that is missing or refuses must not fail the bench line
            return None, "live counter pass %s failed: %s" % (pmc[0], e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    got["dispatches"] = n_disp
    got["seconds"] = time.perf_counter() - t_begin
    return got, ("measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* around `bench.py --inner` (three "
                 "separate passes, %d dispatches of %s each on 2 rotating batches of %d); traffic = FETCH_SIZE x 2 + WRITE_SIZE "
                 "KiB (MI355X_MICROARCH.md: gfx950 tallies 128-B requests at 64 B)" % (n_disp, kernel, batch))


def host_cores():
    """(hardware threads, physical cores) of this host."""
    threads = os.cpu_count() or 1
    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or threads
    except Exception:
        phys = threads
    # a container may see fewer CPUs than the machine has: affinity mask and cgroup CPU quota
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None:
        threads = max(1, min(threads, int(quota + 0.5)))
    return threads, max(1, min(phys, threads))


def cpu_baseline(batch, seconds_budget=20.0):
    """Time the CPU oracle (plain-C port of the reference path, oracle/) on this host's cores on a bounded
    sample of the same workload; also returns its answers for a parity spot check.

    Round 4: the oracle's QP solver keeps one workspace per thread (no allocation per solve), the
    batch loop hands out chunks of 256 instances, the all-core figure is the best of >= 3 repetitions of the FULL batch
    with one thread per physical core (SMT siblings add nothing to this fp64 loop) and the 1-thread rate is measured on
    the same thread team size 1."""
    from oracle import oracle

    o = oracle.LinearMpcZmp(1.0, 2.0, 0.0625)
    threads, cores = host_cores()
    n_all = batch["x0"].shape[0]
    # single thread: 4096 instances (~0.2 s), best of 3, gives the per-core rate
    n1 = min(4096, n_all)
    rate1 = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        o.plan_batch(batch["x0"][:n1], batch["zlim"][:n1], 0.005, want_jerk=False, nthreads=1)
        rate1 = max(rate1, n1 / (time.perf_counter() - t0))
    # all cores: the whole batch, repeated until ~seconds_budget/2 of wall time is used (at least 3 times)
    o.plan_batch(batch["x0"][:4 * cores], batch["zlim"][:4 * cores], 0.005, want_jerk=False, nthreads=cores)  # spin up the team
    n_mt = n_all
    best, reps, t_begin = float("inf"), 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t_begin < 0.5 * seconds_budget and reps < 50):
        t0 = time.perf_counter()
        r = o.plan_batch(batch["x0"][:n_mt], batch["zlim"][:n_mt], 0.005, want_jerk=False, nthreads=cores)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    # where this host stops scaling (a cgroup may cap CPU time without saying so): rate at 1, 2, 4, ... threads, one
    # repetition of a quarter batch each
    curve = {}
    t_ = 2
    while t_ < cores:
        nq = max(4096, n_all // 4)
        t0 = time.perf_counter()
        o.plan_batch(batch["x0"][:nq], batch["zlim"][:nq], 0.005, want_jerk=False, nthreads=t_)
        curve[str(t_)] = round(nq / (time.perf_counter() - t0))
        t_ *= 4
    return dict(value=n_mt / best, unit="solves/s", cores=cores, threads=cores, host_hardware_threads=threads, kind="port",
                scaling_curve=curve,
                sample="the %d-instance rank-0 batch, best of %d repetitions, OpenMP over instances, one thread per "
                       "physical core (%d); C restatement of the reference path (oracle/), not QLD" % (n_mt, reps, cores),
                value_1thread=rate1, parallel_efficiency=(n_mt / best) / (rate1 * cores)), r["zmp"], n_mt


def distributed_info(dist, dev, world):
    """What the SCALE record can be checked against: the backend, the world size torch.distributed reports, the number of
    DISTINCT GPUs the ranks sit on (PCI bus ids all-gathered through the job's own process group) and RCCL's version."""
    import torch

    if world <= 1 or dist is None:
        return {"backend": None, "world_size": 1, "distinct_gpus": 1}
    prop = torch.cuda.get_device_properties(dev)
    ident = "%s/%s/%s" % (os.uname().nodename, getattr(prop, "pci_bus_id", dev.index), getattr(prop, "uuid", ""))
    ids = [None] * world
    dist.all_gather_object(ids, ident)
    info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "distinct_gpus": len(set(ids))}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-steps", type=int, default=300,
                    help="untimed launches BEFORE the W warm-up steps (the same count on every rank): after the idle "
                         "seconds of set-up the GPU needs some tens of ms of continuous load to reach its sustained clock "
                         "(measured: kernel time 0.75 -> 0.68 ms over the first 20 launches); 0 disables")
    ap.add_argument("--no-host-p50", action="store_true",
                    help="skip the host-to-host p50 loops (scripts/prof_zmp.sh: their launches read pinned host memory "
                         "through the same kernel and would be averaged into its rocprofv3 --stats line)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch instances on EVERY GPU; strong: --batch instances in total, GPU r "
                         "takes the contiguous shard [r B/N, (r+1) B/N) (SURVEY.md 8e)")
    ap.add_argument("--ring", type=int, default=8,
                    help="distinct batches the timed steps rotate through (seeds 20250928 + 8 rank + k, all resident in "
                         "HBM): step k solves batch k mod RING, so no call sees its own past -- the library orders a call by "
                         "the pivot counts of the handle's LAST call of that size, and a repeated batch would be its best "
                         "case (VERDICT r5 weak #2); the repeated-batch rate is reported as history.value_repeated")
    ap.add_argument("--inner", action="store_true", help="(internal) the short run rank 0 starts under rocprofv3 for the live counters")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not run the three rocprofv3 --pmc passes of the headline kernel (about 40 s); the line then "
                         "replays the counters of profiles/ when they belong to this build")
    ap.add_argument("--no-history-leg", action="store_true",
                    help="skip the second measurement on a handle without a history (profiling runs: one kind of launch)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline only: skip the `secondary` entries (configs 3, 4, 5: xy 65536, ddp 4096, srb 32768, 20 timed "
                         "steps each over rotating batches, each with roofline / cpu_baseline / parity) the default one-GPU command appends")
    ap.add_argument("--workload", choices=["zmp", "zmp100", "xy", "ddp", "srb", "walk", "multi", "xywalk", "ism", "z", "ddpzmp"], default="zmp",
                    help="zmp (default) = the headline metric; the others measure the remaining classes with the same "
                         "protocol (bench_secondary.py)")
    args = ap.parse_args()
    args.batch_given = any(a.startswith("--batch") for a in sys.argv[1:])
    args.steps_given = any(a.startswith("--steps") for a in sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # (the modulo and the backend override exist to exercise the multi-rank path on a single-GPU box:
    #  CCC_BENCH_BACKEND=gloo with two ranks sharing cuda:0; on the 8-GPU node both are no-ops)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("CCC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    dinfo = distributed_info(dist, dev, world)
    if args.workload != "zmp":
        import bench_secondary

        args.distributed_info = dinfo

        bench_secondary.run(args, rank, world, local_rank, dist)
        if world > 1:
            dist.destroy_process_group()
        return

    from centroidalcontrolcollection_amd import LinearMpcZmp
    from centroidalcontrolcollection_amd import fixtures as fx

    N, dt, n = 32, 0.0625, args.batch
    mpc = LinearMpcZmp(1.0, 2.0, dt, device=local_rank)
    if args.scaling == "strong" and world > 1:
        # one workload of --batch instances in total (seed of rank 0), contiguous shards (sharding.shard_bounds)
        from centroidalcontrolcollection_amd import sharding

        lo, hi = sharding.shard_bounds(args.batch, world)[rank]
        n = hi - lo
        if args.batch % world:
            raise SystemExit("--scaling strong needs --batch divisible by the number of GPUs (all_gather_into_tensor)")
        ring_host = []
        for k in range(max(1, args.ring)):
            full = fx.make_zmp_batch(args.batch, N, dt, 1.0, seed=20250928 + k)
            ring_host.append({kk: np.ascontiguousarray(v[lo:hi]) for kk, v in full.items()})
    else:
        # RING distinct draws of the workload per rank (seed 20250928 + 8 rank + k), all resident in HBM
        ring_host = [fx.make_zmp_batch(n, N, dt, 1.0, seed=20250928 + 8 * rank + k) for k in range(max(1, args.ring))]
    R = len(ring_host)
    batch = ring_host[0]
    ring = [(torch.from_numpy(b["x0"]).to(dev), torch.from_numpy(b["zlim"]).to(dev)) for b in ring_host]
    x0, zlim = ring[0]
    # two output buffers: the all-gather of step k (RCCL's own stream) overlaps the kernel of step k + 1
    zbuf = [torch.empty((n, 2), dtype=torch.float64, device=dev) for _ in range(2)]
    zmp = zbuf[0]
    status = torch.empty((n, 2), dtype=torch.int32, device=dev)
    gathered = [torch.empty((world * n, 2), dtype=torch.float64, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    stream = torch.cuda.current_stream(dev)
    counter = [0]

    def step(ev=None):
        k = counter[0] & 1
        counter[0] += 1
        if pending[k] is not None:
            pending[k].wait()  # the stream waits until the gather that still reads zbuf[k] is done
            pending[k] = None
        if ev is not None:
            ev[0].record(stream)
        rx0, rzl = ring[(counter[0] - 1) % R]  # step k solves batch k mod RING: never the batch of the call before
        mpc.plan_batch_device(rx0, rzl, 0.005, zbuf[k], None, None, stream)
        if ev is not None:
            ev[1].record(stream)
        if world > 1:
            pending[k] = dist.all_gather_into_tensor(gathered[k], zbuf[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    # one untimed launch per batch of the ring with the status array: pivot statistics / status check over all of them
    n_bad, piv_sum = 0, 0.0
    for rx0, rzl in ring:
        mpc.plan_batch_device(rx0, rzl, 0.005, zmp, None, status, stream)
        torch.cuda.synchronize(dev)
        st = status.cpu().numpy()
        n_bad += int(((st & 0xff) != 0).sum())
        piv_sum += float((st >> 8).sum())
    pivots_per_solve = piv_sum / (n * R)

    # clock ramp (untimed, before the W warm-up steps): see --ramp-steps
    for _ in range(args.ramp_steps):
        step()
    drain()
    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    drain()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    drain()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = np.array([a.elapsed_time(b) for a, b in evs])  # HIP events on the launch stream
    timed_kernel = mpc.last_kernel()  # which kernel the timed launches ran: asked of the library (ccc_zmp_last_kernel)
    if os.environ.get("CCC_BENCH_DEBUG") and rank == 0:
        print("kern_ms", np.round(kern_ms[:24], 3).tolist(), "wall_ms", 1e3 * elapsed,
              "span_ms", evs[0][0].elapsed_time(evs[-1][1]), file=sys.stderr)
    # The timed steps above ROTATE through RING distinct batches: no call sees its own past.  The library orders a call by
    # the pivot counts of the handle's last call of that size (csrc/zmp.hip) -- what a closed-loop
    # caller gets from cycle to cycle, and nothing a caller of unrelated batches can use (the handle notices and drops
    # it).  Two more figures beside `value`: the same rotation on a handle that keeps no history at all
    # (CCC_ZMP_HISTORY=0, read at creation), and ONE batch repeated on a fresh handle -- the schedule's best case.
    no_hist = None
    if world == 1 and not args.no_history_leg:
        def timed(h, rotate):
            for i in range(args.warmup + 20):
                rx0, rzl = ring[i % R] if rotate else ring[0]
                h.plan_batch_device(rx0, rzl, 0.005, zbuf[0], None, None, stream)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(args.steps):
                rx0, rzl = ring[i % R] if rotate else ring[0]
                h.plan_batch_device(rx0, rzl, 0.005, zbuf[0], None, None, stream)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t1, h.last_kernel()

        os.environ["CCC_ZMP_HISTORY"] = "0"
        try:
            mpc0 = LinearMpcZmp(1.0, 2.0, dt, device=local_rank)
        finally:
            del os.environ["CCC_ZMP_HISTORY"]
        e0, k0 = timed(mpc0, True)
        del mpc0
        mpc1 = LinearMpcZmp(1.0, 2.0, dt, device=local_rank)
        e1, k1 = timed(mpc1, False)
        del mpc1
        no_hist = {"value_without_history": n * args.steps / e0, "ms_per_step_without_history": 1e3 * e0 / args.steps,
                   "kernel_without_history": k0,
                   "value_repeated": n * args.steps / e1, "ms_per_step_repeated": 1e3 * e1 / args.steps, "kernel_repeated": k1,
                   "what": "`value` rotates through %d distinct batches on a default handle (no call sees its own past); "
                           "value_without_history = the same rotation on a handle created with CCC_ZMP_HISTORY=0; "
                           "value_repeated = ONE batch repeated on a fresh handle, whose calls then run longest-first by the "
                           "previous call's pivot counts, QPs of like counts paired in a wavefront (the answers do not depend "
                           "on the order: tests/test_zmp_gpu.py) -- the schedule's best case, what the device-side closed loop "
                           "of TestLinearMpcZmp.cpp approaches from cycle to cycle (DESIGN.md section 4)" % R}
    # p50 of the host-to-host call: inputs in host memory -> planned ZMPs back in host memory through
    # ccc_zmp_plan_batch; PCIe-inclusive, never the `value` above
    # ... measured twice: from PINNED host tensors (SURVEY.md 8d's definition of the p50: the kernel reads the inputs
    # and writes the ZMPs in the caller's page-locked memory, no copy) and from pageable numpy arrays (staged chunk by
    # chunk through the handle's pinned buffers)
    h2h, h2h_pageable = [], []
    px0 = torch.from_numpy(batch["x0"]).pin_memory()
    pzl = torch.from_numpy(batch["zlim"]).pin_memory()
    pz = torch.empty((n, 2), dtype=torch.float64).pin_memory()
    pgath = torch.empty((world * n, 2), dtype=torch.float64).pin_memory() if world > 1 else None
    n_h2h = 0 if args.no_host_p50 else 200
    for _ in range(20 if n_h2h else 0):
        mpc.plan_batch_pinned(px0, pzl, 0.005, pz)
    if world > 1:
        dist.barrier()
    for _ in range(n_h2h):
        t1 = time.perf_counter()
        mpc.plan_batch_pinned(px0, pzl, 0.005, pz)
        if world > 1:  # "outputs gathered on every rank": through the device buffers RCCL works on
            zbuf[0].copy_(pz, non_blocking=True)
            dist.all_gather_into_tensor(gathered[0], zbuf[0])
            pgath.copy_(gathered[0], non_blocking=True)
            torch.cuda.synchronize(dev)
        h2h.append(1e3 * (time.perf_counter() - t1))
    zref = torch.empty((n, 2), dtype=torch.float64, device=dev)  # batch 0 of the ring through the device entry
    mpc.plan_batch_device(x0, zlim, 0.005, zref, None, None, stream)
    torch.cuda.synchronize(dev)
    assert not n_h2h or np.array_equal(pz.numpy(), zref.cpu().numpy()), "pinned path differs from the device path"
    if rank == 0 and n_h2h:
        for _ in range(8):
            t1 = time.perf_counter()
            mpc.planOnceBatch(batch["x0"], batch["zlim"], 0.005)
            h2h_pageable.append(1e3 * (time.perf_counter() - t1))
        h2h_pageable = h2h_pageable[2:]
    p50_h2h = float(np.median(h2h)) if h2h else 0.0
    if world > 1:
        t = torch.tensor([p50_h2h], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        p50_h2h = float(t.item())

    # weak-scaling runs on N > 1 GPUs also time BASELINE's literal configuration -- 65536 instances in TOTAL, contiguous
    # shards of batch / N per GPU -- so that one launch of the driver's command yields both curves
    strong = None
    if world > 1 and args.scaling == "weak" and args.batch % world == 0:
        ns = args.batch // world
        ring_s = [(a[:ns].contiguous(), b[:ns].contiguous()) for a, b in ring]  # rotating here too
        sz = [torch.empty((ns, 2), dtype=torch.float64, device=dev) for _ in range(2)]
        sg = [torch.empty((world * ns, 2), dtype=torch.float64, device=dev) for _ in range(2)]
        works = [None, None]

        def sstep(i):
            k = i & 1
            if works[k] is not None:
                works[k].wait()
            sx0, szl = ring_s[i % R]
            mpc.plan_batch_device(sx0, szl, 0.005, sz[k], None, None, stream)
            works[k] = dist.all_gather_into_tensor(sg[k], sz[k], async_op=True)

        for i in range(args.warmup):
            sstep(i)
        for w in works:
            if w is not None:
                w.wait()
        works = [None, None]
        dist.barrier()
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for i in range(args.steps):
            sstep(i)
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize(dev)
        dist.barrier()
        es = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
        dist.all_reduce(es, op=dist.ReduceOp.MAX)
        es = float(es.item())
        strong = {"total_batch": args.batch, "batch_per_gpu": ns, "value": args.batch * args.steps / es,
                  "unit": "solves/s", "ms_per_step": 1e3 * es / args.steps,
                  "what": "strong scaling: the same kernel + all-gather on shards of batch/N (BASELINE's 65536 total)"}
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * n * args.steps / elapsed
        kavg = float(kern_ms.mean()) * 1e-3
        achieved = ALGO_BYTES_PER_SOLVE * n / kavg / 1e9
        tflops = pivots_per_solve * FLOP_PER_PIVOT * n / kavg / 1e12
        vc = valu_counters(n, timed_kernel)
        traffic, traffic_src = measured_traffic(n, timed_kernel)
        live = None
        if world == 1 and not args.inner and not args.no_live_counters:
            live, live_src = live_counters(n, timed_kernel)
            if live is not None:
                traffic = live["FETCH_SIZE"] * 1024 * 2 + live["WRITE_SIZE"] * 1024
                traffic_src = live_src
                vc = {"valu_insts_per_solve": live["SQ_INSTS_VALU"] / n,
                      "valu_issue_frac": live["SQ_INSTS_VALU"] * 4.0 / (1024.0 * live["dispatch_ns"] * 2.4)}
            else:
                traffic_src = "%s; live collection: %s" % (traffic_src, live_src)
        out = {
            "metric": "LinearMpcZmp planOnce() solves/sec (N=32, fp64, inputs resident in HBM)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ramp_steps": args.ramp_steps,  # untimed launches before the warm-up (GPU clock ramp), see --ramp-steps
            "ms_per_step": ms_per_step,
            # median wall time from "inputs resident in pinned host memory" to "planned ZMPs back in pinned
            # host memory on every rank" (max over ranks); PCIe-inclusive, never the `value` above
            "p50_ms": p50_h2h if n_h2h else None,
            "p50_kernel_ms": float(np.median(kern_ms)),
            "p50_pageable_host_ms": float(np.median(h2h_pageable)) if n_h2h else None,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "LinearMpcZmp N=32 (2 s horizon @ 62.5 ms), batch=%d per GPU (%s), random 6-step "
                                   "footstep sequences (SURVEY.md 8d), the timed steps rotating through %d distinct batches" % (
                                       n, "%d in total, contiguous shards" % (world * n)
                                       if args.scaling == "strong" and world > 1 else "weak scaling", R),
                       "batch_per_gpu": n, "ring": R, "horizon_steps": N, "parallelism": "batch-sharded x%d" % world,
                       "collective": "all_gather(zmp)" if world > 1 else "none"},
            # the bound that binds is VALU issue (fp64 vector pipe), not HBM and not MFMA: `achieved/peak/frac` are the
            # HBM figures the contract asks for, `valu` the ones that say how far the kernel is from ITS roofline
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # (PMC counters need rocprofv3 around the process: collected by scripts/prof_zmp.sh -- this command
                         #  under `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, one pass each -- and read back from profiles/, only
                         #  when that summary was made from the kernel sources this library was built from)
                         "traffic_source": traffic_src,
                         "algorithmic_bytes": ALGO_BYTES_PER_SOLVE * n,
                         "kernel": timed_kernel,
                         "kernel_avg_ms": kavg * 1e3,
                         "kernel_avg_ms_what": "HIP events round the library call on its stream: the solve kernel plus, on a "
                                               "handle with a history, the two order_by_count kernels (counting sort of the "
                                               "last call's pivot counts) in front of it -- both in profiles/*_zmp_kernel_stats.csv",
                         "valu": {"achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": tflops / FP64_VECTOR_PEAK_TFLOPS,
                                  "flop_per_solve": pivots_per_solve * FLOP_PER_PIVOT,
                                  "issue_frac": vc["valu_issue_frac"] if vc else None,
                                  "insts_per_pivot_trip": vc["valu_insts_per_solve"] / (0.5 * pivots_per_solve) if vc else None,
                                  "what": "useful fp64 flop = pivots x 2048 (rank-1 update of the 32 x 32 tableau) over "
                                          "the kernel time, against the vector-fp64 peak; issue_frac = VALU wave-"
                                          "instructions x 4 clk / (SIMDs x clk) from the PMC pass in profiles/"},
                         "mfma": {"utilisation": 0.0, "measured": "SQ_INSTS_MFMA = 0, SQ_VALU_MFMA_BUSY_CYCLES = 0 per launch "
                                                                      "(profiles/r02_zmp_mfma_counters.csv)",
                                  "why": "the pivot is a rank-1 update of a register-resident 32 x 32 tableau whose pivot "
                                         "row and column depend on the previous pivot: no GEMM-shaped work to tile "
                                         "(the smallest fp64 MFMA, 4x4x4, would run at 1/4 fill on a rank-1 product)"},
                         "note": "algorithmic bytes = 1088 B/solve x batch: 1 % of HBM, the path is not memory-bound "
                                 "(DESIGN.md section 4)"},
            "pivots_per_solve": pivots_per_solve,
            "unsolved": n_bad,
        }
        if live is not None:
            out["roofline"]["live_counters"] = {k: live[k] for k in sorted(live)}
            out["roofline"]["live_counters"]["what"] = (
                "per dispatch of the headline kernel, averaged over the dispatches of the inner runs; dispatch_ns = the "
                "kernel's duration under counter collection (SQ pass), the clock issue_frac is computed with")
        if no_hist is not None:
            out["history"] = no_hist
        out["distributed"] = dinfo
        if strong is not None:
            out["strong_scaling"] = strong
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (the other ranks would idle meanwhile)
            cb, ref_zmp, n_chk = cpu_baseline(batch)
            out["cpu_baseline"] = cb
            out["parity_max_abs_err"] = float(np.abs(zref.cpu().numpy()[:n_chk] - ref_zmp).max())
    # configs 3, 4, 5 in the driver's record: the default one-GPU command also times LinearMpcXY
    # (config 4), DdpCentroidal (config 3) and DdpSingleRigidBody (config 5's shape) with the same protocol and appends their
    # lines -- value, ms_per_step, steps, roofline (traffic + traffic_source), cpu_baseline (value_1thread), parity
    secondary = None
    if world == 1 and not args.no_secondary and not args.batch_given:
        import bench_secondary

        torch.cuda.empty_cache()
        secondary = []
        for wl, (k_steps, k_warm) in (("xy", (20, 2)), ("ddp", (20, 2)), ("srb", (20, 2)), ("zmp100", (20, 2))):
            secondary.append(bench_secondary.measure(wl, bench_secondary.DEFAULT_BATCH[wl], k_steps, k_warm, rank, world,
                                                     local_rank, dist, cpu=not args.no_cpu_baseline, dinfo=None,
                                                     live=not args.no_live_counters and not args.inner))
            torch.cuda.empty_cache()
    if rank == 0:
        if secondary is not None:
            out["secondary"] = secondary
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
