"""Numpy restatement of the control-limited DDP of oracle/ddp.c with SWITCHES for the solver-internal choices the
reference tree does not pin (nmpc_ddp is external): used to bisect which of them the reference's SRB closed loop
(tests/src/TestDdpSingleRigidBody.cpp:105-175, max_iter = 1 per warm-started cycle) depends on.

Usage: python tests/tools/ddp_variants_proto.py [key=value ...]   (see OPTS)
Not a product path and not the oracle: a development tool kept for the record of DESIGN section 7.
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from centroidalcontrolcollection_amd import fixtures_ddp as fd  # noqa: E402

G = 9.80665

OPTS = dict(
    warm_max_iter=1,
    reg_type=1,            # 1: Vxx + lambda I;  2: Quu + lambda I
    boxqp_iter=100,
    clamp_init=0,          # clamp the initial inputs into the limits before the first rollout
    fail="keep",           # what the solve returns when the backward pass fails: keep = initial inputs
    zero_changed=1,        # the reference test zeroes steps whose input dimension changed
    shift=0,
    verbose=0,
    end_time=3.0,
    alpha_n=11,
    trace_from=-1,
    trace_to=-1,
)


class Srb:
    S = 12

    def __init__(self, mass, dt, w):
        self.mass, self.dt = mass, dt
        self.w_run, self.w_term, self.w_force = np.array(w["run"]), np.array(w["term"]), w["force"]

    def bind(self, prob):
        self.pd = prob["phase_dim"][0]
        self.pv, self.pr = prob["phase_vertex"][0], prob["phase_ridge"][0]
        self.sp = prob["step_phase"][0]
        self.ref = np.concatenate([prob["ref_pos"][0], prob["ref_ori"][0]], axis=1)
        self.I = prob["inertia"][0]
        self.Iinv = np.linalg.inv(self.I)

    def m(self, i):
        return int(self.pd[self.sp[i]])

    @staticmethod
    def euler(ori):
        ca, sa, cb, sb = np.cos(ori[0]), np.sin(ori[0]), np.cos(ori[1]), np.sin(ori[1])
        return np.array([[ca * sb / cb, sb * sa / cb, 1.0], [-sa, ca, 0.0], [ca / cb, sa / cb, 0.0]])

    def f(self, i, x, u):
        m = self.m(i)
        V, R = self.pv[self.sp[i], :m], self.pr[self.sp[i], :m]
        pos, ori, v, w = x[0:3], x[3:6], x[6:9], x[9:12]
        xd = np.zeros(12)
        xd[0:3] = v
        xd[3:6] = self.euler(ori) @ w
        xd[6:9] = [0, 0, -G]
        a = -np.cross(w, self.I @ w)
        if m:
            xd[6:9] += (u[:m, None] * R).sum(0) / self.mass
            a = a + (u[:m, None] * np.cross(V - pos, R)).sum(0)
        xd[9:12] = self.Iinv @ a
        return x + self.dt * xd

    def fd(self, i, x, u):
        m = self.m(i)
        V, R = self.pv[self.sp[i], :m], self.pr[self.sp[i], :m]
        pos, ori, w = x[0:3], x[3:6], x[9:12]
        Fx, Fu = np.zeros((12, 12)), np.zeros((12, m))
        Fx[0:3, 6:9] = np.eye(3)
        Fx[3:6, 9:12] = self.euler(ori)
        w1, w2, w3 = w
        ca, sa, cb, sb = np.cos(ori[0]), np.sin(ori[0]), np.cos(ori[1]), np.sin(ori[1])
        cb2, sb2 = cb * cb, sb * sb
        Fx[3:6, 3] = [-w1 * sa * sb / cb + w2 * sb * ca / cb, -w1 * ca - w2 * sa, -w1 * sa / cb + w2 * ca / cb]
        Fx[3:6, 4] = [w1 * sb2 * ca / cb2 + w1 * ca + w2 * sa * sb2 / cb2 + w2 * sa, 0.0,
                      w1 * sb * ca / cb2 + w2 * sa * sb / cb2]
        I = self.I
        I11, I12, I13, I22, I23, I33 = I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]
        D = np.array([[I12 * w3 - I13 * w2, -I13 * w1 + I22 * w3 - 2 * I23 * w2 - I33 * w3,
                       I12 * w1 + I22 * w2 + 2 * I23 * w3 - I33 * w2],
                      [-I11 * w3 + 2 * I13 * w1 + I23 * w2 + I33 * w3, -I12 * w3 + I23 * w1,
                       -I11 * w1 - I12 * w2 - 2 * I13 * w3 + I33 * w1],
                      [I11 * w2 - 2 * I12 * w1 - I22 * w2 - I23 * w3, I11 * w1 + 2 * I12 * w2 + I13 * w3 - I22 * w1,
                       I13 * w2 - I23 * w1]])
        Fx[9:12, 9:12] = self.Iinv @ D
        if m:
            tf = (u[:m, None] * R).sum(0)
            Fu[6:9] = R.T / self.mass
            Fu[9:12] = self.Iinv @ np.cross(V - pos, R).T
            cm = np.array([[0, -tf[2], tf[1]], [tf[2], 0, -tf[0]], [-tf[1], tf[0], 0]])
            Fx[9:12, 0:3] = self.Iinv @ cm
        Fx = Fx * self.dt + np.eye(12)
        return Fx, Fu * self.dt

    def lrun(self, i, x, u):
        m = self.m(i)
        return 0.5 * np.dot(self.w_run, (x - np.concatenate([self.ref[i], np.zeros(6)])) ** 2) \
            + 0.5 * self.w_force * np.dot(u[:m], u[:m])

    def lterm(self, x):
        N = len(self.sp)
        return 0.5 * np.dot(self.w_term, (x - np.concatenate([self.ref[N], np.zeros(6)])) ** 2)


def box_qp(H, g, lo, hi, x0, max_iter):
    n = len(g)
    x = np.clip(x0, lo, hi)
    value = x @ g + 0.5 * x @ H @ x
    clamped = np.zeros(n, bool)
    result, old = 0, 0.0
    free = ~clamped
    Lf = None
    for it in range(1, max_iter + 1):
        if result:
            break
        if it > 1 and (old - value) < 1e-8 * abs(old):
            result = 4
            break
        old = value
        grad = g + H @ x
        oc = clamped
        clamped = ((x == lo) & (grad > 0)) | ((x == hi) & (grad < 0))
        free = ~clamped
        if clamped.all():
            result = 6
            break
        if it == 1 or (oc != clamped).any():
            try:
                Lf = np.linalg.cholesky(H[np.ix_(free, free)])
            except np.linalg.LinAlgError:
                result = -1
                break
        gn = np.linalg.norm(grad[free])
        if gn < 1e-8:
            result = 5
            break
        gc = g + H @ (x * clamped)
        s = np.zeros(n)
        y = np.linalg.solve(Lf.T, np.linalg.solve(Lf, gc[free]))
        s[free] = -y - x[free]
        sdg = s @ grad
        if sdg >= 0:
            break
        step = 1.0
        while True:
            xc = np.clip(x + step * s, lo, hi)
            vc = xc @ g + 0.5 * xc @ H @ xc
            if not ((vc - old) / (step * sdg) < 0.1):
                break
            step *= 0.6
            if step < 1e-22:
                result = 2
                break
        x, value = xc, vc
    else:
        result = 1
    return x, result, free, Lf


def solve(p, N, x0, u_init, max_iter, o, lo_u=0.0, hi_u=1e6, trace=False):
    S = p.S
    ms = [p.m(i) for i in range(N)]
    lam, dlam = 1e-6, 1.0
    lam_min, lam_max, lam_f, lam_thre = 1e-8, 1e10, 1.6, 1e-7
    u = [np.zeros(ms[i]) if u_init is None else np.array(u_init[i][:ms[i]], dtype=float) for i in range(N)]
    if o["clamp_init"]:
        u = [np.clip(v, lo_u, hi_u) for v in u]
    x = [np.array(x0, dtype=float)]
    for i in range(N):
        x.append(p.f(i, x[i], u[i]))

    def total(xs, us):
        return sum(p.lrun(i, xs[i], us[i]) for i in range(N)) + p.lterm(xs[N])

    cost = total(x, u)
    info = dict(initial_cost=cost, alpha=None, status=0, iters=0)
    need = True
    alphas = 10.0 ** np.linspace(0, -3, 11)[:o["alpha_n"]]
    for it in range(1, max_iter + 1):
        info["iters"] = it
        if need:
            D = [p.fd(i, x[i], u[i]) for i in range(N)]
            need = False
        while True:
            # backward
            refN = np.concatenate([p.ref[N], np.zeros(6)])
            Vx, Vxx = p.w_term * (x[N] - refN), np.diag(p.w_term)
            dV = np.zeros(2)
            ks, Ks = [None] * N, [None] * N
            ok = True
            for i in range(N - 1, -1, -1):
                Fx, Fu = D[i]
                m = ms[i]
                refi = np.concatenate([p.ref[i], np.zeros(6)])
                Lx, Lu = p.w_run * (x[i] - refi), p.w_force * u[i]
                Lxx, Luu = np.diag(p.w_run), p.w_force * np.eye(m)
                Qx, Qu = Lx + Fx.T @ Vx, Lu + Fu.T @ Vx
                Qxx, Qxu, Quu = Lxx + Fx.T @ Vxx @ Fx, Fx.T @ Vxx @ Fu, Luu + Fu.T @ Vxx @ Fu
                Vr = Vxx + (lam * np.eye(S) if o["reg_type"] == 1 else 0)
                Qxur, QuuF = Fx.T @ Vr @ Fu, Luu + Fu.T @ Vr @ Fu
                if o["reg_type"] == 2:
                    QuuF = QuuF + lam * np.eye(m)
                k, K = np.zeros(m), np.zeros((m, S))
                if m:
                    k0 = ks[i + 1] if (i + 1 < N and ms[i + 1] == m) else np.zeros(m)
                    k, rc, free, Lf = box_qp(QuuF, Qu, lo_u - u[i], hi_u - u[i], k0, o["boxqp_iter"])
                    if rc < 1:
                        ok = False
                        break
                    if free.any():
                        K[free] = -np.linalg.solve(Lf.T, np.linalg.solve(Lf, Qxur[:, free].T))
                dV += [k @ Qu, 0.5 * k @ Quu @ k]
                Vx = Qx + K.T @ Quu @ k + K.T @ Qu + Qxu @ k
                Vxx = Qxx + K.T @ Quu @ K + K.T @ Qxu.T + Qxu @ K
                Vxx = 0.5 * (Vxx + Vxx.T)
                ks[i], Ks[i] = k, K
            if ok:
                break
            dlam = max(dlam * lam_f, lam_f)
            lam = max(lam * dlam, lam_min)
            if lam > lam_max:
                info["status"] = -1
                return u, x, cost, info
        g = np.mean([np.max(np.abs(ks[i]) / (np.abs(u[i]) + 1.0)) if ms[i] else 0.0 for i in range(N)])
        if g < 1e-4 and lam < lam_thre:
            info["status"] = 1
            break
        accepted = False
        for a in alphas:
            xc, uc = [x[0]], []
            for i in range(N):
                un = np.clip(u[i] + a * ks[i] + Ks[i] @ (xc[i] - x[i]), lo_u, hi_u) if ms[i] else np.zeros(0)
                uc.append(un)
                xc.append(p.f(i, xc[i], un))
            cc = total(xc, uc)
            actual = cost - cc
            expected = -a * (dV[0] + a * dV[1])
            ratio = actual / expected if expected > 0 else np.sign(actual)
            if trace:
                print("    alpha %.4f cost %.6g -> %.6g expected %.4g ratio %.4g lam %.3g" % (a, cost, cc, expected,
                                                                                            ratio, lam))
            if ratio > 0:
                accepted = True
                break
        if accepted:
            dlam = min(dlam / lam_f, 1 / lam_f)
            lam = lam * dlam * (lam > lam_min)
            x, u, cost = xc, uc, cc
            info["alpha"] = a
            need = True
            if actual < 1e-7:
                info["status"] = 2
                break
        else:
            dlam = max(dlam * lam_f, lam_f)
            lam = max(lam * dlam, lam_min)
            if lam > lam_max:
                info["status"] = -1
                break
    return u, x, cost, info


def closed_loop(o):
    N, dt, mass, sim_dt = 100, 0.03, 100.0, 0.005
    inertia = np.array([40.0, 20.0, 10.0])
    p = Srb(mass, dt, fd.srb_weights())
    sim = fd.CentroidalSim(mass, inertia, sim_dt)
    sim.pos = fd.reference_schedule(0.0)[1].copy()
    u_prev, dims_prev = None, None
    t, cycle = 0.0, 0
    worst = dict(pos=0, ori=0, vel=0, w=0)
    while t < o["end_time"]:
        prob = fd.reference_problem(t, N, dt, 4, 16, (0.1, 0.5), True, np.diag(inertia), fd.srb_ori_ref)
        p.bind(prob)
        dims = np.array([p.m(i) for i in range(N)])
        x0 = np.concatenate([sim.pos, sim.ori[::-1], sim.vel, sim.ang_vel])
        u_init = None
        if u_prev is not None:
            u_init = []
            for i in range(N):
                j = min(i + o["shift"], N - 1)
                ui = u_prev[j]
                if len(ui) != dims[i]:
                    ui = np.zeros(dims[i])
                u_init.append(ui)
        tr = o["trace_from"] <= cycle <= o["trace_to"]
        u, x, cost, info = solve(p, N, x0, u_init, 500 if cycle == 0 else o["warm_max_iter"], o, trace=tr)
        if o["verbose"] and (cycle % o["verbose"] == 0 or tr):
            print("%4d t=%.3f pos %s ori %s v %s | c0 %.5g -> %.5g alpha %s st %d it %d" % (
                cycle, t, np.round(sim.pos, 3), np.round(sim.ori, 3), np.round(sim.vel, 3), info["initial_cost"], cost,
                info["alpha"], info["status"], info["iters"]))
        u_prev, dims_prev = u, dims
        ph = prob["step_phase"][0, 0]
        m0 = prob["phase_dim"][0, ph]
        moment, force = fd.total_wrench(prob["phase_vertex"][0, ph], prob["phase_ridge"][0, ph], u[0][:m0], sim.pos)
        ref = prob["ref_pos"][0, 0]
        ori_ref = prob["ref_ori"][0, 0]
        worst["pos"] = max(worst["pos"], np.linalg.norm(sim.pos - ref))
        worst["ori"] = max(worst["ori"], np.linalg.norm(sim.ori - ori_ref))
        worst["vel"] = max(worst["vel"], np.linalg.norm(sim.vel))
        worst["w"] = max(worst["w"], np.linalg.norm(sim.ang_vel))
        t += sim_dt
        sim.update(force, moment)
        if 1.0 <= t < 1.0 + sim_dt:
            sim.addDisturb((0.05, 0.05, 0.0), np.zeros(3))
        cycle += 1
    ref_end = fd.reference_schedule(t)[1]
    fin = dict(pos=np.linalg.norm(sim.pos - ref_end), ori=np.linalg.norm(sim.ori - fd.srb_ori_ref(t)),
               vel=np.linalg.norm(sim.vel), w=np.linalg.norm(sim.ang_vel))
    return worst, fin


if __name__ == "__main__":
    o = dict(OPTS)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        o[k] = type(OPTS[k])(v) if not isinstance(OPTS[k], str) else v
    worst, fin = closed_loop(o)
    print("worst per-cycle (limits 2, 1, 2, 2):", {k: round(float(v), 4) for k, v in worst.items()})
    print("final (limits 0.1 each):", {k: round(float(v), 4) for k, v in fin.items()})
