import sys, numpy as np
sys.path.insert(0, "/root/repo/scratch"); sys.path.insert(0, "/root/repo")
import xy_proto as P
from centroidalcontrolcollection_amd import fixtures_ddp as fd
N, M, wf, w6 = P.N, P.M, P.wf, P.w6
prob, x0 = fd.make_xy_batch(6, N, P.dt, seed=7)
k = 0
Ad, Bd = P.models(prob, k)
# dense Rt (K+N) x 7N in longdouble
LD = np.longdouble
R = np.zeros((6 * N, 6 * N), dtype=LD)
for s in range(N):
    Phi = np.eye(6, dtype=LD)
    for j in range(s, N):
        if j > s: Phi = Ad[j].astype(LD) @ Phi
        R[6 * j:6 * j + 6, 6 * s:6 * s + 6] = np.sqrt(w6)[:, None] * Phi
Rt = np.zeros((6 * N + N, 7 * N), dtype=LD)
for s in range(N):
    Rt[:6 * N, 7 * s:7 * s + 6] = R[:, 6 * s:6 * s + 6]
    Rt[6 * N + s, 7 * s + 6] = 1
Gam = np.zeros((7 * N, 7 * N), dtype=LD)
for s in range(N):
    for r in range(prob["dim"][k, s]):
        b = np.zeros(7, dtype=LD); b[:6] = Bd[s][:, r]; b[6] = prob["ridge"][k, s, r, 2]
        Gam[7 * s:7 * s + 7, 7 * s:7 * s + 7] += np.outer(b, b)
J = np.diag(np.r_[np.ones(6 * N), np.zeros(N)]).astype(LD)
Mall = wf * J + Rt @ Gam @ Rt.T
# inverse in longdouble via numpy? use Gauss-Jordan
def inv_ld(A):
    n = A.shape[0]; A = A.copy(); I = np.eye(n, dtype=LD)
    for c in range(n):
        p = c + np.argmax(np.abs(A[c:, c]))
        A[[c, p]] = A[[p, c]]; I[[c, p]] = I[[p, c]]
        d = A[c, c]; A[c] /= d; I[c] /= d
        for r in range(n):
            if r != c:
                f = A[r, c]; A[r] -= f * A[c]; I[r] -= f * I[c]
    return I
Minv = inv_ld(Mall)
Qref = (Rt.T @ Minv @ Rt).astype(np.float64)
print("cond M", np.linalg.cond(Mall.astype(np.float64)), "Qref max", np.abs(Qref).max())
np.save("/root/repo/scratch/qref.npy", Qref)
