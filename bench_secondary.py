"""Secondary workloads of bench.py (`--workload xy|ddp|srb|ism|z`): the other rows of SURVEY.md section 8, measured
with the same protocol as the headline (W untimed warm-up steps, K timed steps bracketed by barrier + synchronize, max
over ranks, ONE JSON line with `roofline` and `cpu_baseline`).  A step = one pass of the class's batched planOnce()
over one batch of synthetic instances already resident in HBM; weak scaling (every rank its own batch).
Part of bench.py: the `cpu` callbacks below are its cpu_baseline leg (the only place here that touches oracle/)."""
import json
import os
import sys
import time

import numpy as np
import torch

FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
RING = int(os.environ.get("CCC_BENCH_RING", "8"))  # distinct batches the timed steps of the scheduled workloads (xy, ddp, srb, walk, multi) rotate through: no call
#           sees its own past (the handles order a call by what their LAST call of that size measured)


def _dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _tile(d, n, base):
    k = (n + base - 1) // base
    return {a: np.concatenate([v] * k)[:n] for a, v in d.items()}


def _without_history(env, make):
    """A handle created with the library's history switch `env` off (the switches are read when a handle is created)."""
    old = os.environ.get(env)
    os.environ[env] = "0"
    try:
        return make()
    finally:
        if old is None:
            del os.environ[env]
        else:
            os.environ[env] = old


def _xy_algo(status):
    """Bytes a lane must move (csrc/xy.hip, stage-recursion kernel): inputs and outputs once, the ridge vectors into the
    workspace once (7 doubles per ridge), then per iteration and stage: the backward sweep reads 43 doubles (the clamped
    set's sums, the step's scalars) and writes 69 (feedback, value function), the forward sweep reads 80 + the 16 ridge
    vectors (112) and writes 36 (the next set and its sums); iterations = set changes + 1 of the instances that kernel
    finished."""
    it = (status >> 8).astype(np.float64)
    it = np.where(it <= 16, it + 1.0, 17.0).mean()
    return 20 * (4 + 16 * 3 * 8 * 2 + 16 + 48) + 48 + 128 + 20 * 16 * 7 * 8 + it * 20 * (43 + 69 + 80 + 16 * 7 + 36) * 8


XY_MFMA = ("the block Hessian B'WB (the one genuine dense GEMM of this class: 320 x 120 x 320 per instance at config 4, "
           "24.6 Mflop) is never formed: the QP is solved in stage space by 6 x 6 Riccati recursions over the horizon "
           "(csrc/xy.hip), ~0.1 Mflop per instance and iteration -- forming it on MFMA at this solve rate would need ~80 "
           "TFLOP/s of fp64 for a matrix the solver then still has to factorise (DESIGN.md section 7b); 0 v_mfma in the ISA")


def _xy(n, dev, rank, walking=False):
    from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd
    N, dt, base, M = 20, 0.1, min(n, 2048), 16
    if walking:
        # two separate foot contacts in double support (32 ridges per step), 30 steps: beyond the dual kernel, the
        # stage-recursion kernel with its single-change safeguard rounds
        N, M, base = 30, 32, min(n, 512)
        gen = lambda seed: fd.make_xy_walking_batch(base, N, dt, M=32, seed=seed)  # noqa: E731
    else:
        gen = lambda seed: fd.make_xy_batch(base, N, dt, seed=seed)  # noqa: E731
    # RING distinct draws (seed 20250928 + 8 rank + k), each tiled to the batch; entry 0 is the one the CPU leg checks
    ring = []
    for k in range(1 if walking else RING):
        pk, xk = gen(20250928 + 8 * rank + k)
        pk = _tile(pk, n, base)
        xk = np.concatenate([xk] * ((n + base - 1) // base))[:n]
        if k == 0:
            prob, x0 = pk, xk
        ring.append(({a: _dev(v, dev) for a, v in pk.items()}, _dev(xk, dev)))
    mpc = LinearMpcXY(100.0, dt, N, device=dev.index, max_ridges=M)
    tp, tx0 = ring[0]
    out = torch.zeros((n, M), dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    count = [0]

    def step(stream):
        rp, rx = ring[count[0] % len(ring)]
        count[0] += 1
        mpc.plan_batch_device(rp, rx, out, status=st, stream=stream)

    def rebase(stream):  # the outputs of ring entry 0 (what `cpu` compares with the oracle)
        mpc.plan_batch_device(tp, tx0, out, status=st, stream=stream)

    def other(history, rotate):
        make = lambda: LinearMpcXY(100.0, dt, N, device=dev.index, max_ridges=M)  # noqa: E731
        m0 = make() if history else _without_history("CCC_XY_HISTORY", make)
        o0, c0 = torch.zeros_like(out), [0]

        def f(stream):
            rp, rx = ring[c0[0] % len(ring)] if rotate else ring[0]
            c0[0] += 1
            m0.plan_batch_device(rp, rx, o0, stream=stream)
        return f, (m0, o0)

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or (512 if walking else 2048))
        sub = {a: v[:ns] for a, v in prob.items()}
        o = oracle.LinearMpcXY(100.0, dt, N, M=M)
        t0 = time.perf_counter()
        r = o.plan_batch(sub, x0[:ns], nthreads=cores)
        t = time.perf_counter() - t0
        err = np.abs(out.cpu().numpy()[:ns] - r["u0"]).max() / (np.abs(r["u0"]).max() + 1.0)
        return ns / t, ns, float(err), "max |d force scale| / (1 + max force scale)"

    if walking:
        return dict(name="LinearMpcXY planOnce() solves/sec (N=30, 32 ridge slots, fp64, inputs resident in HBM)", step=step,
                    out=out, status=st, rebase=rebase, ring=len(ring),
                    workload="LinearMpcXY N=30 (3 s horizon @ 100 ms), walking with two foot contacts in double support (32 "
                             "ridges), batch=%d per GPU (beyond BASELINE's configs: src/LinearMpcXY.cpp:69-82)" % n,
                    algo_bytes=N * (4 + M * 3 * 8 * 2 + 16 + 48) + 48 + M * 8, kernel="xy_plan_stream_kernel<32,false>", cpu=cpu,
                    keep=(mpc, tp, tx0), mfma=XY_MFMA)
    return dict(name="LinearMpcXY planOnce() solves/sec (N=20, fp64, inputs resident in HBM)", step=step, out=out, status=st,
                workload="LinearMpcXY N=20 (2 s horizon @ 100 ms), 16 ridges per step, batch=%d per GPU (BASELINE config 4)" % n,
                algo_bytes=20 * (4 + 16 * 3 * 8 * 2 + 16 + 48) + 48 + 128, stream_bytes=_xy_algo,
                kernel="xy_plan_stream_kernel<16,false>", cpu=cpu, keep=(mpc, ring), mfma=XY_MFMA, other=other,
                # LinearMpcXY is held to 5e-9 of the force scales relative to (1 + the largest), the bound of the certified
                # golden vectors (tests/test_xy_gpu.py): condition number 1e6-1e7, two different exact solvers
                parity_tol=5e-9,
                rebase=rebase, ring=len(ring),
                history_what="the first round of the block iteration takes the instances in the order of the sweeps the "
                             "handle's last call of this size spent on each (DESIGN.md section 8)")


def _ddp(n, dev, rank, srb, walking=False):
    from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd
    N, dt, base = (50, 0.03, min(n, 4096)) if srb else (100, 0.03, min(n, 4096))
    kw, M, P = {}, 16, 4
    if walking == "multi":
        # feet + hands: 32- / 48- / 64-ridge steps -> the tile kernel with four ridge blocks
        N, dt, base, M = 30, 0.05, min(n, 512), 64
        gen = lambda seed: fd.make_multicontact_batch(base, N, dt, seed=seed, srb=srb)  # noqa: E731
        P = gen(20250928)[0]["phase_dim"].shape[1]
        kw = dict(max_phases=P, max_ridges=64)
    elif walking:
        # double-support walking sequences: 32-ridge steps, 8-10 contact phases -> the tile kernel with two ridge blocks
        N, dt, base, M = 40, 0.05, min(n, 1024), 32
        gen = lambda seed: fd.make_walking_batch(base, N, dt, seed=seed, srb=srb)  # noqa: E731
        P = gen(20250928)[0]["phase_dim"].shape[1]
        kw = dict(max_phases=P, max_ridges=32)
    else:
        gen = lambda seed: fd.make_centroidal_batch(base, N, dt, seed=seed, srb=srb)  # noqa: E731
    # RING distinct draws (seed 20250928 + 8 rank + k), each tiled to the batch; entry 0 is the one the CPU leg checks
    ring_host = []
    for k in range(RING):
        pk, xk = gen(20250928 + 8 * rank + k)
        if pk["phase_dim"].shape[1] != P:  # (the walking fixtures' phase count follows the draw: pad to the handle's)
            pad = P - pk["phase_dim"].shape[1]
            if pad < 0:
                continue
            pk = dict(pk, phase_dim=np.pad(pk["phase_dim"], ((0, 0), (0, pad))),
                      phase_vertex=np.pad(pk["phase_vertex"], ((0, 0), (0, pad), (0, 0), (0, 0))),
                      phase_ridge=np.pad(pk["phase_ridge"], ((0, 0), (0, pad), (0, 0), (0, 0))))
        ring_host.append((_tile(pk, n, base), np.concatenate([xk] * ((n + base - 1) // base))[:n]))
    prob, x0 = ring_host[0]
    def handle():
        if srb:
            w = DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3, terminal_pos=(1.0, 1.0, 10.0),
                                               terminal_ori=(0.5,) * 3)
            h = DdpSingleRigidBody(100.0, dt, N, w, device=dev.index, **kw)
        else:
            h = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)),
                              device=dev.index, **kw)
        h.ddp_solver_.config().max_iter = 20
        return h

    d = handle()
    ring = [({a: _dev(v, dev) for a, v in pk.items()}, _dev(xk, dev)) for pk, xk in ring_host]
    tp, tx0 = ring[0]
    out = torch.zeros((n, N, M), dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)
    count = [0]

    def step(stream):
        rp, rx = ring[count[0] % len(ring)]
        count[0] += 1
        d.plan_batch_device(rp, rx, out, iters=it, status=st, stream=stream)

    def rebase(stream):  # the outputs of ring entry 0 (what `cpu` compares with the oracle)
        d.plan_batch_device(tp, tx0, out, iters=it, status=st, stream=stream)

    def other(history, rotate):
        d0 = handle() if history else _without_history("CCC_DDP_HISTORY", handle)
        o0, c0 = torch.zeros_like(out), [0]

        def f(stream):
            rp, rx = ring[c0[0] % len(ring)] if rotate else ring[0]
            c0[0] += 1
            d0.plan_batch_device(rp, rx, o0, stream=stream)
        return f, (d0, o0)

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or 2048)
        sub = {a: v[:ns] for a, v in prob.items()}
        o = oracle.Ddp(1 if srb else 0, 100.0, dt, N, fd.srb_weights() if srb else fd.centroidal_weights(), max_iter=20,
                       P=P, M=M, arith=d.arithmetic())
        t0 = time.perf_counter()
        r = o.plan_batch(sub, x0[:ns], nthreads=cores)
        t = time.perf_counter() - t0
        same = bool(np.array_equal(out.cpu().numpy()[:ns], r["u"]))
        return ns / t, ns, 0.0 if same else float(np.abs(out.cpu().numpy()[:ns] - r["u"]).max()), \
            "max |d force scale| over the whole planned sequence (0.0 = bit-identical)"

    S = 12 if srb else 9
    return dict(name="%s planOnce() solves/sec (horizon %d, <= 20 DDP iterations, %s, inputs resident in HBM)"
                % ("DdpSingleRigidBody" if srb else "DdpCentroidal", N,
                   "fp64"),
                step=step, out=out, status=st, iters=it, dtype="f64",
                workload=("%s horizon=%d @ %d ms, max_iter=20, batch=%d per GPU (%s)"
                          % ("DdpSingleRigidBody" if srb else "DdpCentroidal", N, round(dt * 1e3), n,
                             ("feet + hands multi-contact, 32 / 48 / 64 ridges per step, %d contact phases: beyond BASELINE's configs, "
                              "src/DdpCentroidal.cpp:49-60" % P) if walking == "multi" else
                             "walking with 32-ridge double support, %d contact phases: beyond BASELINE's configs, "
                             "src/DdpCentroidal.cpp:49-60" % P if walking else
                             "BASELINE config %s" % ("5 (fp64: its fp32 request runs this kernel, DESIGN.md 7.5)" if srb else "3"))),
                algo_bytes=P * 4 + 2 * P * M * 3 * 8 + N * 4 + (N + 1) * 24 * (2 if srb else 1) + (72 if srb else 0)
                + S * 8 + N * M * 8,
                kernel="ddp_tile_kernel<%d, %d>" % (S, M // 16), cpu=cpu,
                valu=lambda iters: _ddp_valu(S, M, N, iters, walking),
                keep=(d, ring), other=other, rebase=rebase, ring=len(ring), parity_tol=0.0,  # (bit-identical to the tile oracle)
                history_what="fresh instances are handed out longest-first from the busy times of the handle's last call of "
                             "this size, and none is suspended (DESIGN.md section 7.4)")


def _ddp_valu(S, M, N, iters, walking):
    """What the DDP kernel is bound by -- VALU instruction throughput -- from the newest committed PMC summary made from the
    same kernel sources (profiles/r*_ddp_valu_counters.json; replayed, see replayed_counters), plus a DENSE-EQUIVALENT flop
    count for orientation: SURVEY.md 8(d)'s count per backward step of the dense formulation (Quu 2(S^2 m + S m^2), Qxu /
    Qxx 2(S^3 + S^2 m), Cholesky m^3 / 3, gains 2 m^2 S, value update 6 k at S = 9, m = 16: ~25 kflop) x horizon x the
    iterations the run executed.  The structured backward step (round 4) gets the same result with fewer operations, so
    this is NOT what the kernel executes and is kept out of any fraction of the peak."""
    import glob

    from centroidalcontrolcollection_amd import build as _b

    m = M
    per_step = 2 * (S * S * m + S * m * m) + 2 * (S ** 3 + S * S * m) + m ** 3 / 3 + 2 * m * m * S + 6e3 * (S / 9.0) ** 2
    pmc, src = None, None
    here = os.path.dirname(os.path.abspath(__file__))
    mine = _b.kernel_hash("ddp")
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_ddp_valu_counters.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        rel = "profiles/" + os.path.basename(path)
        if d.get("kernel_hash") != mine:
            src = src or "%s refused: profiled build %s, this build %s" % (rel, d.get("kernel_hash"), mine)
            continue
        pmc = d.get("S%d" % S if M == 16 else "S%dM%d" % (S, M))
        src = "%s (replayed; kernel_hash %s = this build)" % (rel, mine)
        break
    return dict(dense_equivalent_flop_per_solve=per_step * N * iters, dense_equivalent_flop_per_backward_step=per_step,
                issue_frac=None if pmc is None else pmc["valu_issue_frac"],
                wait_frac=None if pmc is None else pmc["wait_any_frac"],
                simd_valu_busy_frac=None if pmc is None else pmc.get("simd_valu_busy_frac"),
                arch_vgprs=None if pmc is None else pmc.get("arch_vgpr", pmc.get("vgpr")),
                scratch_bytes_per_lane=None if pmc is None else pmc.get("scratch"),
                counters_source=src,
                what="simd_valu_busy_frac = SQ_ACTIVE_INST_VALU x wavefronts per SIMD / SQ_WAVE_CYCLES (clock-independent: how "
                     "busy a SIMD's one vector ALU is -- the roofline this kernel is bound by); issue_frac = SQ_INSTS_VALU x 4 "
                     "clk / (SIMDs x kernel clocks at the 2.4 GHz PEAK clock); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                     "dense_equivalent_* = flop of the DENSE formulation of the backward passes the run executed (the "
                     "structured kernel performs fewer): orientation only, in no fraction of the peak")


def _zmp100(n, dev, rank):
    """LinearMpcZmp at the reference test's own horizon (TestLinearMpcZmp.cpp:17-19: 2 s @ 20 ms = 100 steps): the
    state-space kernel KS (csrc/zmp_stage.inc) with the exact dual active set behind it.  RING distinct batches."""
    from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
    N, dt, base = 100, 0.02, min(n, 4096)
    # RING distinct draws (seed 20250928 + 8 rank + k), each tiled to the batch (KS keeps no schedule: a repeated instance
    # costs what a fresh one does); entry 0 is the one the CPU leg checks
    bs = [_tile(fx.make_zmp_batch(base, N, dt, seed=20250928 + 8 * rank + k), n, base) for k in range(RING)]
    mpc = LinearMpcZmp(1.0, 2.0, dt, device=dev.index)
    ring = [(_dev(b["x0"], dev), _dev(b["zlim"], dev)) for b in bs]
    out = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    st = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    turn = [0]

    def step(stream):
        x0, zl = ring[turn[0] % RING]
        turn[0] += 1
        mpc.plan_batch_device(x0, zl, 0.005, out, None, st, stream=stream)

    def rebase(stream):
        mpc.plan_batch_device(ring[0][0], ring[0][1], 0.005, out, None, st, stream=stream)

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or 4096)
        o = oracle.LinearMpcZmp(1.0, 2.0, dt)
        t0 = time.perf_counter()
        r = o.plan_batch(bs[0]["x0"][:ns], bs[0]["zlim"][:ns], 0.005, want_jerk=False, nthreads=cores)
        t = time.perf_counter() - t0
        return ns / t, ns, float(np.abs(out.cpu().numpy()[:ns] - r["zmp"]).max()), "max |d ZMP| [m]"

    def valu_replay():
        """The state-space kernel is bound by its own vector instructions on one wavefront per SIMD: the share of the SIMD's
        VALU cycles it keeps busy, from the newest committed PMC summary made from the same kernel sources (replayed)."""
        import glob

        from centroidalcontrolcollection_amd import build as _b

        here, mine, src = os.path.dirname(os.path.abspath(__file__)), _b.kernel_hash("zmp100"), None
        for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_zmp100_valu_counters.json")), reverse=True):
            with open(path) as f:
                d = json.load(f)
            rel = "profiles/" + os.path.basename(path)
            if d.get("kernel_hash") != mine:
                src = src or "%s refused: profiled build %s, this build %s" % (rel, d.get("kernel_hash"), mine)
                continue
            return dict(simd_valu_busy_frac=d["simd_valu_busy_frac"], wait_frac=d["wait_frac"],
                        valu_insts_per_wavefront=d["valu_insts_per_wavefront"],
                        salu_insts_per_wavefront=d["salu_insts_per_wavefront"], effective_clock_ghz=d["effective_clock_ghz"],
                        counters_source="%s (replayed; kernel_hash %s = this build)" % (rel, mine),
                        what="simd_valu_busy_frac = SQ_ACTIVE_INST_VALU x wavefronts per SIMD / SQ_WAVE_CYCLES of "
                             "zmp_plan_stage_kernel: one wavefront per SIMD, which issues its scalar bookkeeping and waits for "
                             "its loads itself (DESIGN.md 4.1; scripts/fp64_rate_probe.hip: one wavefront with four independent "
                             "fp64 chains reaches 59 of the 62 TFLOP/s the part sustains)")
        return dict(simd_valu_busy_frac=None, counters_source=src)

    C = 8  # (stages per checkpoint, csrc/zmp.hip CCC_ZMP_STAGE_CHUNK)
    npad, nc = (N + C - 1) // C * C, (N + C - 1) // C

    def stream_bytes(status):  # per instance (two QPs): limits transposed once, then per iteration the limits twice + checkpoints
        it = (status >> 8).astype(np.float64).reshape(-1)
        it = np.where(it <= 20, it, 20.0)  # (a handed-over QP carries the exact kernel's pivot count: KS spent its limit on it)
        pad = (-len(it)) % 64
        wave = np.concatenate([it, np.zeros(pad)]).reshape(-1, 64).max(axis=1) + 0.5  # a wavefront sweeps until its last lane
        per_qp = 2 * npad * 16 + wave * (npad * 40 + nc * 144)                        # is done (+ the costate pass over the
        return float(per_qp.sum() * 64 / len(it) * 2)                                 # stored jerks); per instance: 2 QPs

    return dict(name="LinearMpcZmp planOnce() solves/sec (N=100, the reference test's horizon, fp64, inputs resident in HBM)",
                step=step, rebase=rebase, out=out, status=st, ring=RING,
                workload="LinearMpcZmp N=100 (2 s horizon @ 20 ms, TestLinearMpcZmp.cpp:17-19), random 6-step footstep "
                         "sequences, batch=%d per GPU" % n,
                algo_bytes=2 * (24 + 2 * N * 8) + 16, stream_bytes=stream_bytes, kernel="zmp_plan_stage_kernel", cpu=cpu,
                keep=(mpc, ring), parity_tol=1e-9, valu_replay=valu_replay)


def _ism(n, dev, rank):
    from centroidalcontrolcollection_amd import IntrinsicallyStableMpc, fixtures as fx
    N, dt, base = 100, 0.02, min(n, 1024)
    b = _tile(fx.make_ism_batch(base, N, dt, seed=20250928 + rank), n, base)
    mpc = IntrinsicallyStableMpc(1.0, 2.0, dt, device=dev.index)
    ti, tr = _dev(b["init"], dev), _dev(b["ref"], dev)
    out = torch.zeros((n, 2), dtype=torch.float64, device=dev)
    st = torch.zeros((n, 2), dtype=torch.int32, device=dev)

    def step(stream):
        mpc.plan_batch_device(ti, tr, 0.005, out, status=st, stream=stream)

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or 16384)
        o = oracle.IntrinsicallyStableMpc(1.0, 2.0, dt)
        t0 = time.perf_counter()
        r = o.plan_batch(b["init"][:ns], b["ref"][:ns], 0.005, want_vel=False, nthreads=cores)
        t = time.perf_counter() - t0
        return ns / t, ns, float(np.abs(out.cpu().numpy()[:ns] - r["zmp"]).max()), "max |d ZMP| [m]"

    return dict(name="IntrinsicallyStableMpc planOnce() solves/sec (N=100, fp64, inputs resident in HBM)", step=step, out=out,
                status=st, workload="IntrinsicallyStableMpc N=100 (2 s horizon @ 20 ms), batch=%d per GPU" % n,
                algo_bytes=2 * (16 + 3 * N * 8) + 16, kernel="ism_plan_pcr_kernel", cpu=cpu, keep=(mpc, ti, tr))


def _z(n, dev, rank):
    from centroidalcontrolcollection_amd import LinearMpcZ, fixtures as fx
    N, dt, base = 40, 0.05, min(n, 4096)
    b = _tile(fx.make_z_batch(base, N, dt, seed=20250928 + rank), n, base)
    mpc = LinearMpcZ(100.0, dt, N, device=dev.index)
    tc, tr, tx = _dev(b["contact"], dev), _dev(b["ref_pos"], dev), _dev(b["x0"], dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)

    def step(stream):
        mpc.plan_batch_device(tc, tr, tx, out, status=st, stream=stream)

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or 65536)
        o = oracle.LinearMpcZ(100.0, dt, N)
        t0 = time.perf_counter()
        r = o.plan_batch(b["contact"][:ns], b["ref_pos"][:ns], b["x0"][:ns], nthreads=cores)
        t = time.perf_counter() - t0
        err = (np.abs(out.cpu().numpy()[:ns] - r["force"]) / (np.abs(r["force"]) + 1.0)).max()
        return ns / t, ns, float(err), "max relative |d force|"

    def algo(status):
        # bytes a lane must move (csrc/z.hip, streaming kernel): inputs and outputs, the reference and the start forces into
        # the workspace, then per Newton iteration one backward sweep (ref, f, flag in; flag, three gains out) and one
        # forward sweep (f, flag, gains, ref in; f out) over the horizon; `it` = iterations at convergence
        it = float((status >> 8).mean())
        return N * 4 + N * 8 + 16 + 8 + N * 16 + (it + 1.0) * N * (8 + 8 + 1 + 1 + 24) + it * N * (8 + 1 + 24 + 8 + 8)

    return dict(name="LinearMpcZ planOnce() solves/sec (N=40, fp64, inputs resident in HBM)", step=step, out=out, status=st,
                workload="LinearMpcZ N=40 (2 s horizon @ 50 ms), batch=%d per GPU" % n,
                algo_bytes=N * 4 + N * 8 + 16 + 8, stream_bytes=algo,
                kernel="z_plan_stream_kernel", cpu=cpu, keep=(mpc, tc, tr, tx))


def _ddpzmp(n, dev, rank):
    from centroidalcontrolcollection_amd import DdpZmp, fixtures as fx
    N, dt, base, max_iter = 100, 0.02, min(n, 2048), 3
    b = _tile(fx.make_ddpzmp_batch(base, N, dt, seed=20250928 + rank), n, base)
    d = DdpZmp(100.0, dt, N, device=dev.index)
    d.ddp_solver_.config().max_iter = max_iter  # tests/src/TestDdpZmp.cpp:29
    tr, tx, tu = _dev(b["ref"], dev), _dev(b["x0"], dev), _dev(b["u_init"], dev)
    u = torch.zeros((n, N, 3), dtype=torch.float64, device=dev)
    it = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    out = torch.zeros((n, 3), dtype=torch.float64, device=dev)  # PlannedData (zmp, force_z) = the first input

    def step(stream):
        d.plan_batch_device(tr, tx, tu, u, None, it, st, None, stream=stream)
        out.copy_(u[:, 0, :])

    def cpu(cores, ns=None):
        from oracle import oracle
        ns = min(n, ns or 65536)
        o = oracle.DdpZmp(100.0, dt, N, max_iter=max_iter)
        t0 = time.perf_counter()
        r = o.plan_batch(b["ref"][:ns], b["x0"][:ns], b["u_init"][:ns], nthreads=cores)
        t = time.perf_counter() - t0
        err = np.abs(u.cpu().numpy()[:ns] - r["u"]).max()
        return ns / t, ns, float(err), "max |d u| over the whole planned input sequence (0 = bit-identical)"

    # bytes a lane must move: RefData + warm start in, per iteration one backward (13 in, 21 out) and one forward (34 in,
    # 9 out) per horizon step through the workspace, the planned inputs out
    algo = 8 * ((N + 1) * 4 + N * 3 + 6 + max_iter * N * (13 + 21 + 34 + 9) + N * 3)
    return dict(name="DdpZmp planOnce() solves/sec (horizon 100, 3 DDP iterations, fp64, inputs resident in HBM)", step=step,
                out=out, status=st, iters=it,
                workload="DdpZmp horizon=100 @ 20 ms, max_iter=3, warm start (TestDdpZmp.cpp:17-29), batch=%d per GPU" % n,
                algo_bytes=8 * ((N + 1) * 4 + N * 3 + 6 + N * 3), stream_bytes=algo, kernel="ddpzmp_plan_kernel", cpu=cpu,
                keep=(d, tr, tx, tu, u))


DEFAULT_BATCH = dict(zmp100=32768, xy=65536, ddp=4096, srb=32768, ism=65536, z=65536, ddpzmp=65536, walk=4096, multi=2048, xywalk=32768)
DEFAULT_STEPS = dict(zmp100=(20, 3), xy=(5, 1), ddp=(3, 1), srb=(2, 1), ism=(20, 3), z=(50, 5), ddpzmp=(20, 3),
                     walk=(3, 1), multi=(3, 1), xywalk=(3, 1))


# sample of the 1-thread leg of cpu_baseline (instances; ~1 s each on one core of the GPU box's host)
ONE_THREAD_SAMPLE = dict(zmp100=512, xy=64, xywalk=16, ddp=128, srb=256, walk=96, multi=64, ism=1024, z=8192, ddpzmp=4096)


def replayed_counters(workload, n):
    """(hbm_bytes_per_step | None, source): the measured HBM bytes of one step of `workload` at batch n from the newest
    committed PMC summary (profiles/r*_hbm_traffic.json, written by scripts/summarize_round.py from `rocprofv3 --pmc
    FETCH_SIZE` / `WRITE_SIZE` passes around this very command).  A counter needs rocprofv3 around the process, so the
    figure is REPLAYED, not measured in this run -- and only when the summary was made from the same kernel sources
    (centroidalcontrolcollection_amd.build.kernel_hash: csrc files + flags) as the library that just ran; otherwise
    null, and the source string says which build the summary belongs to."""
    import glob

    from centroidalcontrolcollection_amd import build as _b

    here = os.path.dirname(os.path.abspath(__file__))
    mine = _b.kernel_hash(workload)
    refused = None
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                tr = json.load(f).get(workload)
        except Exception:
            continue
        if not tr or tr.get("batch") != n:
            continue
        rel = "profiles/" + os.path.basename(path)
        if tr.get("kernel_hash") != mine:
            refused = refused or "%s refused: profiled build %s, this build %s" % (rel, tr.get("kernel_hash"), mine)
            continue
        return tr.get("hbm_bytes_per_launch", tr.get("hbm_bytes_per_step")), "%s (replayed; kernel_hash %s = this build)" % (rel, mine)
    return None, refused


def run(args, rank, world, local_rank, dist):
    n = args.batch if args.batch_given else DEFAULT_BATCH[args.workload]
    strong = getattr(args, "scaling", "weak") == "strong" and world > 1
    if strong:
        # BASELINE's literal configurations (config 4: 65536 in total over 8 GPUs, config 5: 32768): contiguous shards
        if n % world:
            raise SystemExit("--scaling strong needs the batch divisible by the number of GPUs")
        n //= world
    steps, warmup = (args.steps, args.warmup) if args.steps_given else DEFAULT_STEPS[args.workload]
    out = measure(args.workload, n, steps, warmup, rank, world, local_rank, dist, strong=strong,
                  cpu=not args.no_cpu_baseline, dinfo=getattr(args, "distributed_info", None),
                  history_leg=not getattr(args, "no_history_leg", False),
                  live=not getattr(args, "no_live_counters", False) and not getattr(args, "inner", False))
    if out is not None:
        print(json.dumps(out))


def live_secondary(workload, n, kernel):
    """The counters of a secondary workload measured in this run: `bench.py --workload W --inner` (one
    batch, 1 + 3 + 1 calls) under `rocprofv3 --pmc`, FETCH_SIZE and WRITE_SIZE passes and, for the DDP kernels, an SQ pass.
    HBM bytes per step = (FETCH_SIZE x 2 + WRITE_SIZE) KiB summed over EVERY dispatch of the library's kernels in the run /
    its calls.  Returns (dict | None, source)."""
    import bench

    calls = 1 + 3 + 1  # warm-up, timed steps, the closing call that leaves ring entry 0's answers
    here = os.path.dirname(os.path.abspath(__file__))
    inner = [sys.executable, os.path.join(here, "bench.py"), "--workload", workload, "--inner", "--batch", str(n), "--steps", "3",
             "--warmup", "1", "--no-cpu-baseline", "--no-history-leg"]
    passes = [["FETCH_SIZE"], ["WRITE_SIZE"]]
    ddp = "ddp_tile" in kernel
    if ddp:
        passes.append(["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"])
    t0 = time.perf_counter()
    got, why = bench.pmc_passes(inner, passes, env={"CCC_BENCH_RING": "1"})
    if got is None:
        return None, why
    mine = lambda kn: "ccc_amd::" in kn  # noqa: E731  (not the runtime's copy / fill kernels)
    tot = {c: sum(v[0] for kn, v in got.get(c, {}).items() if mine(kn)) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    res = {"hbm_bytes_per_step": (tot["FETCH_SIZE"] * 2 + tot["WRITE_SIZE"]) * 1024 / calls, "calls": calls,
           "seconds": time.perf_counter() - t0}
    if ddp:
        want = kernel.replace(" ", "")
        sq = {c: [v for kn, v in got.get(c, {}).items() if want in kn.replace(" ", "")] for c in passes[2]}
        if all(len(v) == 1 for v in sq.values()):
            a = {c: v[0][0] / v[0][1] for c, v in sq.items()}  # per dispatch
            res.update(simd_valu_busy_frac=a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"] * a["SQ_WAVES"] / 1024.0,
                       wait_frac=a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], valu_insts_per_instance=a["SQ_INSTS_VALU"] / n)
    return res, ("measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE%s around `bench.py --workload %s --inner` "
                 "(separate passes, %d calls of one batch of %d); bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB over every dispatch of "
                 "the library's kernels / calls" % (" | SQ_*" if ddp else "", workload, calls, n))


def measure(workload, n, steps, warmup, rank, world, local_rank, dist, strong=False, cpu=True, dinfo=None, history_leg=True,
            live=False):
    """One bench line (a dict; None on ranks other than 0) of a secondary workload: `warmup` untimed steps, `steps` timed
    ones bracketed by barrier + synchronize, max over ranks.  Also what bench.py's default command appends to the headline
    line as `secondary` (configs 3, 4, 5: VERDICT r4 item 2)."""
    dev = torch.device("cuda", local_rank)
    make = dict(zmp100=_zmp100, xy=_xy, xywalk=lambda a, b, c: _xy(a, b, c, True), ism=_ism, z=_z, ddpzmp=_ddpzmp, ddp=lambda a, b, c: _ddp(a, b, c, False), srb=lambda a, b, c: _ddp(a, b, c, True),
                walk=lambda a, b, c: _ddp(a, b, c, False, True), multi=lambda a, b, c: _ddp(a, b, c, False, "multi"))
    w = make[workload](n, dev, rank)
    stream = torch.cuda.current_stream(dev)
    gathered = (torch.empty((world * w["out"].shape[0],) + tuple(w["out"].shape[1:]), dtype=w["out"].dtype, device=dev)
                if world > 1 else None)

    def step(ev=None):
        if ev is not None:
            ev[0].record(stream)
        w["step"](stream)
        if ev is not None:
            ev[1].record(stream)
        if world > 1:
            dist.all_gather_into_tensor(gathered, w["out"])

    for _ in range(warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        step(evs[k])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = np.array([a.elapsed_time(b) for a, b in evs])
    if "rebase" in w:  # (the timed steps rotated through the ring: leave ring entry 0's answers in the output arrays)
        w["rebase"](stream)
        torch.cuda.synchronize(dev)
    if rank != 0:
        return None
    kavg = float(kern_ms.mean()) * 1e-3
    st = w["status"].cpu().numpy()
    # ALGORITHMIC bytes = the mandatory inputs + outputs of an instance, nothing else.  The kernels that
    # stream their per-instance state through an HBM workspace (one instance per lane) also report that stream --
    # modelled bytes per instance from the iteration counts of the run -- as `workspace`: a TRAFFIC figure (how close the
    # streaming is to the HBM peak), not a roofline fraction.
    sb = w.get("stream_bytes")
    if callable(sb):
        sb = sb(st)
    achieved = w["algo_bytes"] * n / kavg / 1e9
    # measured HBM bytes per step of THIS workload at THIS batch, replayed from the newest PMC summary made from the same
    # kernel sources (replayed_counters above); null otherwise
    traffic, traffic_src = replayed_counters(workload, n)
    live_res = None
    if live and world == 1 and workload in ("xy", "ddp", "srb", "zmp100"):
        live_res, live_src = live_secondary(workload, n, w["kernel"])
        if live_res is not None:
            traffic, traffic_src = live_res["hbm_bytes_per_step"], live_src
        else:
            traffic_src = "%s; live collection: %s" % (traffic_src, live_src)
    out = {"metric": w["name"], "value": world * n * steps / elapsed, "unit": "solves/s", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "p50_ms": float(np.median(kern_ms)),
           "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": w.get("dtype", "f64"), "data": "synthetic",
           "distributed": dinfo,
           "config": {"workload": w["workload"] + (", the timed steps rotating through %d distinct batches" % w["ring"]
                                                   if w.get("ring", 1) > 1 else ""),
                      "batch_per_gpu": n, "total_batch": world * n, "ring": w.get("ring", 1),
                      "parallelism": "batch-sharded x%d" % world,
                      "collective": "all_gather(planned outputs)" if world > 1 else "none"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "algorithmic_bytes": w["algo_bytes"] * n,
                        "kernel": w["kernel"], "kernel_avg_ms": kavg * 1e3,
                        "workspace": None if sb is None else {
                            "modelled_bytes": sb * n, "achieved": sb * n / kavg / 1e9, "unit": "GB/s",
                            "frac_of_hbm_peak": sb * n / kavg / 1e9 / HBM_PEAK_GBS,
                            "what": "inputs + outputs + the per-iteration state this kernel streams through its HBM "
                                    "workspace (model from the iteration counts; the measured counter traffic is in "
                                    "profiles/): a traffic figure, not the roofline fraction"},
                        "mfma": {"utilisation": 0.0,
                                 "why": w.get("mfma", "no GEMM-shaped work: per-instance fp64 recursions over 6 x 6 .. 32 x 32 "
                                              "blocks with data-dependent pivots / active sets, rank-1 updates and "
                                              "triangular solves on dependent chains (DESIGN.md per class)")},
                        "note": "algorithmic bytes = mandatory inputs + outputs per instance x batch (SURVEY.md 8d); "
                                "none of these kernels is bound by that stream (DESIGN.md says what binds each)"},
           "unsolved": int((st < 0).sum()) if "iters" in w else int(((st & 0xff) != 0).sum())}  # DDP: status < 0 = failure,
    #                                      0 = iteration limit, 1 / 2 = converged (oracle/ddp.c); QPs: low byte != 0
    if "iters" in w:
        out["mean_iterations"] = float(w["iters"].float().mean().item())
    if "valu" in w:
        # the DDP planners are bound by the latency of dependent fp64 VALU / LDS operations, not by HBM: the figures that
        # say how far the kernel is from THAT roofline (useful flop from SURVEY.md 8(d)'s count per backward step x the
        # steps the run executed, against the vector-fp64 peak; the issue share from the PMC pass in profiles/)
        v = w["valu"](out["mean_iterations"])
        out["roofline"]["bound"] = "valu"
        if live_res is not None and "simd_valu_busy_frac" in live_res:  # (this run's SQ pass instead of the replayed summary)
            v = dict(v, simd_valu_busy_frac=live_res["simd_valu_busy_frac"], wait_frac=live_res["wait_frac"],
                     valu_insts_per_instance=live_res["valu_insts_per_instance"], counters_source=live_src)
        out["roofline"]["valu"] = dict(achieved=v["simd_valu_busy_frac"], peak=1.0, unit="share of SIMD VALU cycles busy",
                                       frac=v["simd_valu_busy_frac"],
                                       dense_equivalent_tflops=v["dense_equivalent_flop_per_solve"] * n / kavg / 1e12, **v)
    if "valu_replay" in w:
        v = w["valu_replay"]()
        if v.get("simd_valu_busy_frac") is not None:
            out["roofline"]["bound"] = "valu"
        out["roofline"]["valu"] = dict(achieved=v.get("simd_valu_busy_frac"), peak=1.0, unit="share of SIMD VALU cycles busy",
                                       frac=v.get("simd_valu_busy_frac"), **v)
    if world == 1 and history_leg and "other" in w:
        # `value` rotates through w["ring"] distinct batches on a default handle.  Beside it: the same rotation on a handle
        # that keeps no history at all, and ONE batch repeated on a fresh handle (the schedule's best case)
        def timed(history, rotate):
            step0, keep0 = w["other"](history, rotate)
            for _ in range(max(2, warmup)):
                step0(stream)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(steps):
                step0(stream)
            torch.cuda.synchronize(dev)
            e = time.perf_counter() - t1
            del step0, keep0
            return e

        e0, e1 = timed(False, True), timed(True, False)
        out["history"] = {"value_without_history": n * steps / e0, "ms_per_step_without_history": 1e3 * e0 / steps,
                          "value_repeated": n * steps / e1, "ms_per_step_repeated": 1e3 * e1 / steps,
                          "what": "`value`: the timed steps rotate through %d distinct batches on a default handle (no call "
                                  "sees its own past); value_without_history: the same rotation on a handle created with the "
                                  "history switched off; value_repeated: ONE batch repeated on a fresh handle -- " % w["ring"]
                                  + w["history_what"] + ".  The answers are the same bits whatever the schedule (tests/)"}
    if cpu and world == 1:
        import bench  # (host_cores: physical cores within the affinity mask and the cgroup CPU quota)

        hw_threads, cores = bench.host_cores()
        rate, ns, err, what = w["cpu"](cores)
        rate1, ns1, _, _ = w["cpu"](1, ONE_THREAD_SAMPLE[workload])
        out["cpu_baseline"] = {"value": rate, "unit": "solves/s", "cores": cores, "threads": cores,
                               "host_hardware_threads": hw_threads, "kind": "port", "value_1thread": rate1,
                               "parallel_efficiency": rate / (rate1 * cores),
                               "sample": "first %d instances of the rank-0 batch, OpenMP over instances, one thread per "
                                         "physical core of the cgroup quota (%d of the host's %d hardware threads); 1-thread "
                                         "rate on the first %d; C restatement of the reference path (oracle/), not the "
                                         "reference's Eigen + QLD / nmpc_ddp build" % (ns, cores, hw_threads, ns1)}
        out["parity"] = {"value": err, "what": what}
        if "parity_tol" in w:  # (the tolerance this class is held to, stated and asserted in the run)
            out["parity"]["tolerance"] = w["parity_tol"]
            assert err <= w["parity_tol"], "%s: parity %g exceeds the stated tolerance %g" % (workload, err, w["parity_tol"])
    return out
