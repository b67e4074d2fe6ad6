"""Prototype (numpy; not used by product or tests): a block primal-dual active set on the headline QP of LinearMpcZmp
(KKT system (G mu)_i = lo_i / hi_i / in between, G = B_seq B_seq^T) instead of the Goldfarb-Idnani dual iteration of
csrc/zmp_k1.inc -- measured, round 4, 4000 QPs of the bench workload: 45.4 set changes (= tableau sweeps) and 8.4 iterations
per QP against 15.8 pivots of the dual iteration (|W| at the optimum: 15.2), 0.85 % of the QPs cycle.  Not pursued.
usage: python tests/tools/zmp_pdas_proto.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from centroidalcontrolcollection_amd import fixtures as fx
from oracle import oracle
N, dt = 32, 0.0625
n = 2000
b = fx.make_zmp_batch(n, N, dt, seed=1)
o = oracle.LinearMpcZmp(1.0, 2.0, dt)
A, B = o.seq()
G = B @ B.T
ref = o.plan_batch(b["x0"], b["zlim"], 0.005)
print("oracle iters mean", ref["iters"].mean() if "iters" in ref else None)
def gi_count(lo, hi):
    # reference: number of clamped at optimum via scipy-free PDAS converged set
    pass
tot_changes=[]; its=[]; fails=0; nW=[]
for k in range(n):
    for ax in range(2):
        x0 = b["x0"][k, ax]; fr = A @ x0
        lo = b["zlim"][k, ax, 0] - fr; hi = b["zlim"][k, ax, 1] - fr
        # KKT: z = G mu; z_i = lo_i if mu_i>0 ; hi_i if mu_i<0; lo<=z<=hi if mu_i=0
        state = np.zeros(N, int)   # +1 at lo (mu>0), -1 at hi (mu<0)
        changes = 0; conv=False; seen=set()
        for it in range(40):
            W = state != 0
            mu = np.zeros(N)
            if W.any():
                d = np.where(state > 0, lo, hi)[W]
                mu[W] = np.linalg.solve(G[np.ix_(W, W)], d)
            z = G @ mu
            new = state.copy()
            # release wrong-sign multipliers
            new[(state > 0) & (mu <= 0)] = 0
            new[(state < 0) & (mu >= 0)] = 0
            # clamp violated
            free = state == 0
            new[free & (z < lo - 1e-12)] = 1
            new[free & (z > hi + 1e-12)] = -1
            if np.array_equal(new, state): conv=True; break
            key = new.tobytes()
            if key in seen: break
            seen.add(key)
            changes += int((new != state).sum()); state = new
        if not conv: fails += 1
        tot_changes.append(changes); its.append(it+1); nW.append(int((state!=0).sum()))
print("PDAS: fails %d / %d; iterations mean %.2f max %d; set changes (sweeps) mean %.1f; |W| final mean %.1f" % (fails, 2*n, np.mean(its), np.max(its), np.mean(tot_changes), np.mean(nW)))
