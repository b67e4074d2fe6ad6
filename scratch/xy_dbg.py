import sys, numpy as np
sys.path.insert(0, "/root/repo/scratch"); sys.path.insert(0, "/root/repo")
import xy_proto as P
from centroidalcontrolcollection_amd import fixtures_ddp as fd
N, M, wf, w6 = P.N, P.M, P.wf, P.w6
prob, x0s = fd.make_xy_batch(6, N, P.dt, seed=7)
k = 0; x0 = x0s[k]
Ad, Bd = P.models(prob, k)
nv = N * M
# dense Bhat
Bh = np.zeros((6 * N, nv)); Ah = np.zeros((6 * N, 6))
for s in range(N):
    Phi = np.eye(6)
    for j in range(s, N):
        if j > s: Phi = Ad[j] @ Phi
        Bh[6 * j:6 * j + 6, s * M:(s + 1) * M] = Phi @ Bd[s]
Phi = np.eye(6)
for j in range(N):
    Phi = Ad[j] @ Phi; Ah[6 * j:6 * j + 6] = Phi
W = np.tile(w6, N)
H = Bh.T @ (W[:, None] * Bh) + wf * np.eye(nv)
g = -Bh.T @ (W * (prob["ref_out"][k].reshape(-1) - Ah @ x0))
A = np.zeros((N, nv))
for s in range(N):
    A[s, s * M:(s + 1) * M] = prob["ridge"][k, s, :, 2]
b = prob["total_force_z"][k]
# equality-constrained minimiser, all free
K = np.block([[H, A.T], [A, np.zeros((N, N))]])
sol = np.linalg.solve(K, np.r_[-g, b])
lam0 = sol[:nv]
# first direction for p
import xy_proto
Q = None
lam, lam_pre, npiv, stat = None, None, None, None
# run structured init only: replicate by calling with pivot loop disabled
src = open("xy_proto.py").read()
