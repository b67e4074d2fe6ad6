"""Second prototype for the next LinearMpcXY kernel: the primal-dual active set of xy_pdas_proto.py
with its equality-constrained solve done stage by stage -- inputs and the stage equality eliminated in closed form
(input cost w_f I => 6 x 6 algebra only), backward recursion for the value function, forward pass for states, costates,
stage multipliers, force scales and bound multipliers.  numpy only; not used by product or tests.
usage: python tests/tools/xy_pdas_riccati_proto.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroidalcontrolcollection_amd import fixtures_ddp as fd
from oracle import oracle as orc
from xy_stage_space_proto import models, N, dt, mass, M, G, w6, wf, LO, HI

def solve_set(Ad, Bd, az, dim, fz, ref, x0, state):
    """state [N][M]: 0 free, -1 at LO, +1 at HI.  Returns lam [N][M], mult [N][M]."""
    W = np.diag(w6)
    P = np.zeros((6, 6)); p = np.zeros(6)
    E = np.zeros((N, 6, 6)); f = np.zeros((N, 6)); Pt = np.zeros((N, 6, 6)); pt = np.zeros((N, 6))
    tt = np.zeros((N, 6)); al = np.zeros(N); dd = np.zeros(N)
    for s in range(N - 1, -1, -1):
        m = dim[s]
        Pt[s] = P + W; pt[s] = p - W @ ref[s]
        S = np.zeros((6, 6)); t = np.zeros(6); alpha = 0.0; c = np.zeros(6); dprime = fz[s]
        for r in range(m):
            b = Bd[s][:, r]
            if state[s, r] == 0:
                S += np.outer(b, b); t += b * az[s, r]; alpha += az[s, r] ** 2
            else:
                v = LO if state[s, r] < 0 else HI
                c += b * v; dprime -= az[s, r] * v
        if m > 0 and alpha > 0:
            Sp = S - np.outer(t, t) / alpha; cp = c + t * dprime / alpha
        else:
            Sp = S; cp = c
        Mm = np.eye(6) + Sp @ Pt[s] / wf
        E[s] = np.linalg.solve(Mm, Ad[s]); f[s] = np.linalg.solve(Mm, cp - Sp @ pt[s] / wf)
        tt[s] = t; al[s] = alpha; dd[s] = dprime
        P = Ad[s].T @ Pt[s] @ E[s]; P = 0.5 * (P + P.T); p = Ad[s].T @ (Pt[s] @ f[s] + pt[s])
    lam = np.zeros((N, M)); mult = np.zeros((N, M)); x = x0.copy()
    for s in range(N):
        y = E[s] @ x + f[s]; pi = Pt[s] @ y + pt[s]
        nu = -(wf * dd[s] + tt[s] @ pi) / al[s] if al[s] > 0 else 0.0
        for r in range(dim[s]):
            b = Bd[s][:, r]
            if state[s, r] == 0:
                lam[s, r] = -(b @ pi + nu * az[s, r]) / wf
            else:
                lam[s, r] = LO if state[s, r] < 0 else HI
                mult[s, r] = wf * lam[s, r] + b @ pi + nu * az[s, r]
        x = y
    return lam, mult

def pdas(prob, k, x0, maxit=40):
    Ad, Bd = models(prob, k)
    dim = prob["dim"][k]; az = prob["ridge"][k, :, :, 2]; fz = prob["total_force_z"][k]; ref = prob["ref_out"][k]
    state = np.zeros((N, M), int); hist = set()
    for it in range(maxit):
        lam, mult = solve_set(Ad, Bd, az, dim, fz, ref, x0, state)
        new = state.copy()
        valid = np.arange(M)[None, :] < dim[:, None]
        new[valid & (state == 0) & (lam < LO)] = -1
        new[valid & (state == 0) & (lam > HI)] = +1
        new[(state < 0) & (mult < 0)] = 0
        new[(state > 0) & (mult > 0)] = 0
        # a stage must keep at least one free variable for its equality
        for s in range(N):
            if dim[s] > 0 and not (new[s, :dim[s]] == 0).any():
                new[s, np.argmin(np.abs(mult[s, :dim[s]]))] = 0
        if np.array_equal(new, state): return lam, it + 1, True
        key = new.tobytes()
        if key in hist: return lam, it + 1, False
        hist.add(key); state = new
    return lam, maxit, False

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    prob, x0 = fd.make_xy_batch(n, N, dt, seed=3)
    o = orc.LinearMpcXY(mass, dt, N).plan_batch(prob, x0, nthreads=8, want_all=True)
    its, ok, worst, worst0 = [], 0, 0.0, 0.0
    for k in range(n):
        lam, it, conv = pdas(prob, k, x0[k])
        its.append(it); ok += conv
        dim = prob["dim"][k]
        comp = np.concatenate([lam[s, :dim[s]] for s in range(N)])
        ref = o["lam"][k][:len(comp)]
        if conv:
            worst = max(worst, np.abs(comp - ref).max() / (1 + np.abs(ref).max()))
            worst0 = max(worst0, np.abs(lam[0, :dim[0]] - o["u0"][k][:dim[0]]).max() / (1 + np.abs(o["u0"][k]).max()))
    print("PDAS + stage recursion: converged %d / %d, iterations mean %.1f max %d, worst rel err all %.2e, step 0 %.2e"
          % (ok, n, np.mean(its), np.max(its), worst, worst0))
