"""Prototype for the next LinearMpcXY kernel: a primal-dual active-set / semismooth-Newton iteration on
the QP of src/LinearMpcXY.cpp:116-182 -- guess the clamped set, solve the equality-constrained QP on the free set (in a
kernel: one stage-wise Riccati sweep; here: dense), move the set by the signs of the bound multipliers and the bound
violations -- against the oracle, to count iterations.  Dense linear algebra, numpy only; not used by product or tests.
usage: python tests/tools/xy_pdas_proto.py [n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroidalcontrolcollection_amd import fixtures_ddp as fd
from oracle import oracle as orc
from xy_stage_space_proto import models, N, dt, mass, M, G, w6, wf, LO, HI

def build_qp(prob, k, x0):
    Ad, Bd = models(prob, k)
    dim = prob["dim"][k]
    idx = [(s, r) for s in range(N) for r in range(dim[s])]
    nv = len(idx)
    # condensed: x_{j+1} = Phi(j+1,0) x0 + sum_s<=j Phi(j+1,s+1) Bd[s] lam_s
    Bh = np.zeros((6 * N, nv)); free = np.zeros(6 * N)
    x = x0.copy()
    for j in range(N):
        x = Ad[j] @ x; free[6 * j:6 * j + 6] = x
    for c, (s, r) in enumerate(idx):
        v = Bd[s][:, r].copy()
        for j in range(s, N):
            if j > s: v = Ad[j] @ v
            Bh[6 * j:6 * j + 6, c] = v
    W = np.tile(w6, N)
    H = Bh.T @ (W[:, None] * Bh) + wf * np.eye(nv)
    g = -Bh.T @ (W * (prob["ref_out"][k].reshape(-1) - free))
    steps = [s for s in range(N) if dim[s] > 0]
    A = np.zeros((len(steps), nv)); d = np.zeros(len(steps))
    for a, s in enumerate(steps):
        for c, (ss, r) in enumerate(idx):
            if ss == s: A[a, c] = prob["ridge"][k, s, r, 2]
        d[a] = prob["total_force_z"][k, s]
    return H, g, A, d, idx

def pdas(H, g, A, d, maxit=60):
    nv = len(g)
    state = np.zeros(nv, int)   # 0 free, -1 at LO, +1 at HI
    hist = set()
    for it in range(maxit):
        F = state == 0
        lam = np.where(state < 0, LO, np.where(state > 0, HI, 0.0))
        nF = int(F.sum()); ne = A.shape[0]
        K = np.zeros((nF + ne, nF + ne)); K[:nF, :nF] = H[np.ix_(F, F)]; K[:nF, nF:] = A[:, F].T; K[nF:, :nF] = A[:, F]
        rhs = np.concatenate([-g[F] - H[np.ix_(F, ~F)] @ lam[~F], d - A[:, ~F] @ lam[~F]])
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        lam[F] = sol[:nF]; nu = sol[nF:]
        mult = H @ lam + g + A.T @ nu          # bound multipliers on the clamped ones (gradient of the Lagrangian)
        new = state.copy()
        new[F & (lam < LO)] = -1; new[F & (lam > HI)] = +1
        new[(state < 0) & (mult < 0)] = 0      # at LO the Lagrangian gradient must be >= 0
        new[(state > 0) & (mult > 0)] = 0
        if np.array_equal(new, state): return lam, it + 1, True
        key = new.tobytes()
        if key in hist: return lam, it + 1, False   # cycling
        hist.add(key); state = new
    return lam, maxit, False

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    prob, x0 = fd.make_xy_batch(n, N, dt, seed=3)
    o = orc.LinearMpcXY(mass, dt, N).plan_batch(prob, x0, nthreads=8, want_all=True)
    its, ok, worst = [], 0, 0.0
    for k in range(n):
        H, g, A, d, idx = build_qp(prob, k, x0[k])
        lam, it, conv = pdas(H, g, A, d)
        its.append(it); ok += conv
        ref = o["lam"][k][:len(idx)]
        if conv: worst = max(worst, np.abs(lam - ref).max() / (1 + np.abs(ref).max()))
    print("PDAS: converged %d / %d, iterations mean %.1f max %d, worst rel err vs oracle %.2e ; oracle GI iterations mean %.1f"
          % (ok, n, np.mean(its), np.max(its), worst, o["iters"].mean()))
