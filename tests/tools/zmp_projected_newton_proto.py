"""Prototype kept as evidence: projected Newton with a Riccati sweep per step for LinearMpcZmp in the
space of the ZMP outputs.  Gradient and Newton step are right (checked against a dense solve), but the bounds sit on
outputs, not inputs, and the iteration needs 30-400 sweeps where the dual active set of csrc/zmp.hip needs ~18 pivots.
Not used by the product or the tests.  usage: python tests/tools/zmp_projected_newton_proto.py"""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from centroidalcontrolcollection_amd import fixtures as fx
from oracle import oracle
G = 9.80665
def solve(x0, lo, hi, N, dt, h, maxit=60):
    Ad = np.array([[1, dt, dt*dt/2], [0, 1, dt], [0, 0, 1.0]]); Bd = np.array([dt**3/6, dt*dt/2, dt]); C = np.array([1.0, 0.0, -h/G])
    ca = Ad.T @ C; b0 = C @ Bd
    F = Ad - np.outer(Bd, ca) / b0; Gv = Bd / b0
    # start: w = clamp(free response output)  (u = 0 would give w_i = ca x_i)
    def rollout_w(w):
        x = x0.copy(); us = np.zeros(N)
        for i in range(N):
            us[i] = (w[i] - ca @ x) / b0
            x = Ad @ x + Bd * us[i]
        return 0.5 * us @ us, us, x
    x = x0.copy(); w = np.zeros(N)
    for i in range(N):   # greedy start: u = 0 unless the bound forces otherwise
        wi = np.clip(ca @ x, lo[i], hi[i]); w[i] = wi
        u = (wi - ca @ x) / b0; x = Ad @ x + Bd * u
    prev = None; alpha = 1.0; sweeps = 1
    for it in range(maxit):
        J, us, xN = rollout_w(w)   # (in the kernel: carried over from the forward sweep)
        # backward
        mu = np.zeros(3); P = np.zeros((3, 3)); p = np.zeros(3)
        K = np.zeros((N, 3)); k = np.zeros(N); free = np.zeros(N, bool)
        xs = [x0.copy()]
        for i in range(N): xs.append(Ad @ xs[-1] + Bd * us[i])
        sweeps += 1
        for i in range(N - 1, -1, -1):
            grad = us[i] / b0 + Gv @ mu
            cl = (w[i] <= lo[i] and grad > 0) or (w[i] >= hi[i] and grad < 0)
            free[i] = not cl
            if free[i]:
                quu = 1.0 + Bd @ P @ Bd; qux = Bd @ P @ Ad; qu = Bd @ p
                K[i] = -qux / quu; k[i] = -qu / quu
                Pn = Ad.T @ P @ Ad - np.outer(qux, qux) / quu; pn = Ad.T @ p - qux * qu / quu
            else:
                Kc = -ca / b0; kc = w[i] / b0
                Pn = np.outer(Kc, Kc) + F.T @ P @ F; pn = Kc * kc + F.T @ (P @ Gv * w[i] + p)
            P, p = Pn, pn
            mu = F.T @ mu + us[i] * (-ca / b0)
        if prev is not None and np.array_equal(prev, free) and alpha == 1.0:
            return w, us, it, sweeps
        prev = free.copy()
        alpha = 1.0
        while True:
            sweeps += 1
            xn = x0.copy(); xc = x0.copy(); wc = np.zeros(N); Jc = 0.0
            for i in range(N):
                if free[i]:
                    un = K[i] @ xn + k[i]
                else:
                    un = (w[i] - ca @ xn) / b0
                xn = Ad @ xn + Bd * un
                wn = C @ xn
                wc[i] = np.clip(w[i] + alpha * (wn - w[i]), lo[i], hi[i])
                uc = (wc[i] - ca @ xc) / b0
                xc = Ad @ xc + Bd * uc; Jc += 0.5 * uc * uc
            if Jc <= J * (1 + 1e-12) + 1e-300 or alpha < 1e-6: break
            alpha *= 0.5
        if Jc > J * (1 + 1e-12) + 1e-300: pass
        w = wc
        if alpha != 1.0: alpha_used = alpha
        # note alpha stays as used
    return w, us, maxit, sweeps
N, dt, h = 32, 0.0625, 1.0
b = fx.make_zmp_batch(300, N, dt, seed=20250928)
o = oracle.LinearMpcZmp(1.0, 2.0, dt)
ref = o.plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
worst = 0; its = []; sw = []
for kk in range(300):
    for ax in range(2):
        w, us, it, sweeps = solve(b["x0"][kk, ax], b["zlim"][kk, ax, 0], b["zlim"][kk, ax, 1], N, dt, h)
        uo = ref["jerk"][kk, ax]
        err = np.abs(us - uo).max() / max(1.0, np.abs(uo).max())
        worst = max(worst, err); its.append(it); sw.append(sweeps)
sw = np.array(sw)
print("worst rel jerk err", worst, "iterations mean/max", np.mean(its), np.max(its), "sweeps mean", sw.mean(), "hist", np.bincount(np.minimum(sw, 30)))
print("oracle pivots mean", ref["iters"].mean() if "iters" in ref else None)
