"""Generate tests/golden/{xy,ism,z}_golden.npz: known answers for CCC::LinearMpcXY, CCC::IntrinsicallyStableMpc and
CCC::LinearMpcZ from an INDEPENDENT construction and solver.

The reference holds no golden vectors for these paths and cannot be built here.  Each QP is
strictly convex, so its minimiser is unique and a point that satisfies the KKT conditions IS the answer -- whoever
computed it.  This script therefore
  * builds the QP data from the reference's formulas with numpy / scipy only (scipy.linalg.expm for the ZOH of
    include/CCC/StateSpaceModel.h:164-216, the condensed matrices by SIMULATING the discrete dynamics instead of the
    condensing loops of VariantSequentialExtension.h:110-208 that the oracle restates) -- nothing of oracle/ or of the
    HIP path is involved;
  * solves it with a textbook primal active-set method on dense KKT systems (Nocedal & Wright, Alg. 16.3), then
    polishes the solution on the final active set in long double (Gaussian elimination with partial pivoting written
    here, iterative refinement with long-double residuals);
  * stores the KKT certificate next to the answer: stationarity, primal feasibility and the signs of the bound / row
    multipliers, all checked below before anything is written.

Run:  python tests/golden/make_golden_qp.py [xy|xy_wide|ism|z ...]   (needs scipy; writes *_golden.npz next to this file;
      xy_wide -- 700-variable QPs in long double -- takes several minutes and is not part of the default set)
"""
import os
import sys

import numpy as np
from scipy.linalg import expm

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from centroidalcontrolcollection_amd import fixtures as fx  # noqa: E402  (input generators only)
from centroidalcontrolcollection_amd import fixtures_ddp as fd  # noqa: E402

G = 9.80665
LD = np.longdouble
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------ linear algebra
def solve_ld(A, b):
    """Gaussian elimination with partial pivoting in long double."""
    A = np.array(A, dtype=LD)
    b = np.array(b, dtype=LD)
    n = len(b)
    for k in range(n):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if p != k:
            A[[k, p]] = A[[p, k]]
            b[[k, p]] = b[[p, k]]
        f = A[k + 1:, k] / A[k, k]
        A[k + 1:, k:] -= f[:, None] * A[k, k:][None, :]
        b[k + 1:] -= f * b[k]
    x = np.zeros(n, dtype=LD)
    for k in range(n - 1, -1, -1):
        x[k] = (b[k] - A[k, k + 1:] @ x[k + 1:]) / A[k, k]
    return x


def active_set_qp(H, g, A, b, lo, hi, x, max_iter=5000, mult_tol=1e-9):
    """min 1/2 x'Hx + g'x, A x = b, lo <= x <= hi from a feasible x: primal active set, one change per iteration.
    Returns (x, stat) with stat = -1 / 0 / +1 for variables at the lower bound / free / at the upper bound."""
    n, me = len(g), len(b)
    stat = np.where(x <= lo, -1, np.where(x >= hi, 1, 0))
    for it in range(max_iter):
        free = stat == 0
        nf = int(free.sum())
        grad = H @ x + g
        # step on the free variables with A_f p = 0
        K = np.zeros((nf + me, nf + me))
        K[:nf, :nf] = H[np.ix_(free, free)]
        K[:nf, nf:] = A[:, free].T
        K[nf:, :nf] = A[:, free]
        rhs = np.concatenate([-grad[free], np.zeros(me)])
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0] if me and nf < me else np.linalg.solve(K, rhs)
        p = np.zeros(n)
        p[free] = sol[:nf]
        nu = sol[nf:]
        # a zero step up to the accuracy of the float64 KKT solve (the long-double polish and its certificate decide)
        if np.abs(p).max(initial=0.0) <= 1e-7 * max(1.0, np.abs(x).max()) or -(grad @ p) <= 1e-13 * max(1.0, abs(g @ x)):
            mu = grad + A.T @ nu  # multipliers of the bounds: must be >= 0 at lo, <= 0 at hi
            tol = mult_tol * max(1.0, np.abs(mu).max())
            bad = np.where(((stat == -1) & (mu < -tol)) | ((stat == 1) & (mu > tol)))[0]
            if len(bad) == 0:
                return x, stat
            worst = bad[np.argmax(np.abs(mu[bad]))]
            stat[worst] = 0
            continue
        alpha, block = 1.0, -1
        for i in np.where(free)[0]:
            if p[i] < 0 and x[i] + alpha * p[i] < lo[i]:
                alpha, block = (lo[i] - x[i]) / p[i], i
            if p[i] > 0 and x[i] + alpha * p[i] > hi[i]:
                alpha, block = (hi[i] - x[i]) / p[i], i
        x = x + alpha * p
        if block >= 0:
            x[block] = lo[block] if p[block] < 0 else hi[block]
            stat[block] = -1 if p[block] < 0 else 1
    raise RuntimeError("active set did not terminate")


def polish(H, g, A, b, lo, hi, stat):
    """Solve the equality-constrained QP of the active set `stat` in long double; returns (x, nu, mu, certificate)."""
    n, me = len(g), len(b)
    free = stat == 0
    x = np.where(stat == -1, lo, np.where(stat == 1, hi, 0.0)).astype(LD)
    Hl, gl, Al, bl = H.astype(LD), g.astype(LD), A.astype(LD), b.astype(LD)
    nf = int(free.sum())
    K = np.zeros((nf + me, nf + me), dtype=LD)
    K[:nf, :nf] = Hl[np.ix_(free, free)]
    K[:nf, nf:] = Al[:, free].T
    K[nf:, :nf] = Al[:, free]
    rhs = np.concatenate([-(gl[free] + Hl[np.ix_(free, ~free)] @ x[~free]), bl - Al[:, ~free] @ x[~free]])
    sol = solve_ld(K, rhs)
    for _ in range(2):
        sol = sol + solve_ld(K, rhs - K @ sol)
    x[free] = sol[:nf]
    nu = sol[nf:]
    mu = Hl @ x + gl + Al.T @ nu
    cert = dict(stationarity=float(np.abs(mu[free]).max(initial=0.0)),
                equality=float(np.abs(Al @ x - bl).max(initial=0.0)),
                bound_violation=float(max((lo - x).max(), (x - hi).max(), 0.0)),
                multiplier_sign=float(max((-mu[stat == -1]).max(initial=0.0), (mu[stat == 1]).max(initial=0.0))))
    return x.astype(np.float64), nu.astype(np.float64), mu.astype(np.float64), cert


def solve_certified(H, g, A, b, lo, hi, x_feasible, scale, mult_tol=1e-9, sign_tol=1e-9):
    x, stat = active_set_qp(H, g, A, b, lo, hi, x_feasible.copy(), mult_tol=mult_tol)
    x, nu, mu, cert = polish(H, g, A, b, lo, hi, stat)
    assert cert["stationarity"] <= 1e-9 * scale and cert["equality"] <= 1e-9 * scale, cert
    assert cert["bound_violation"] <= 1e-9 * scale and cert["multiplier_sign"] <= sign_tol * scale, cert
    return x, stat, cert


def zoh(A, B, E, dt):
    """Ad, Bd, Ed of xdot = A x + B u + E by one matrix exponential (StateSpaceModel.h:182-214)."""
    n, m = A.shape[0], B.shape[1]
    M = np.zeros((n + m + 1, n + m + 1))
    M[:n, :n] = dt * A
    M[:n, n:n + m] = dt * B
    M[:n, n + m] = dt * E
    X = expm(M)
    return X[:n, :n], X[:n, n:n + m], X[:n, n + m]


# ------------------------------------------------------------------------------------------------ LinearMpcXY
def xy_qp(prob, k, x0, mass, dt, w_out=(1.0, 0.0, 1.0, 0.0, 1.0, 1.0), w_force=1e-5):
    """QP of src/LinearMpcXY.cpp:116-182 for instance k: (H, g, A, b, lo, hi, dims)."""
    N = prob["dim"].shape[1]
    dims = prob["dim"][k]
    models = []
    for i in range(N):
        m = int(dims[i])
        A = np.zeros((6, 6))
        fz, cz = prob["total_force_z"][k, i], prob["com_z"][k, i]
        A[0, 1] = 1
        A[2, 3] = 1
        A[4, 2] = -fz / mass
        A[5, 0] = fz / mass
        B = np.zeros((6, m))
        for r in range(m):
            p, rho = prob["vertex"][k, i, r], prob["ridge"][k, i, r]
            B[:, r] = [0, rho[0], 0, rho[1], -(p[2] - cz) * rho[1] + p[1] * rho[2], (p[2] - cz) * rho[0] - p[0] * rho[2]]
        Ad, Bd, _ = zoh(A, B, np.zeros(6), dt)
        models.append((Ad, Bd))
    n = int(dims.sum())
    off = np.concatenate([[0], np.cumsum(dims)])

    def simulate(xs, u):
        out = np.zeros(6 * N)
        x = xs.copy()
        for i, (Ad, Bd) in enumerate(models):
            x = Ad @ x + (Bd @ u[off[i]:off[i + 1]] if dims[i] else 0.0)
            out[6 * i:6 * i + 6] = x
        return out

    free_resp = simulate(x0, np.zeros(n))
    Bs = np.zeros((6 * N, n))
    for j in range(n):
        e = np.zeros(n)
        e[j] = 1.0
        Bs[:, j] = simulate(np.zeros(6), e)
    W = np.tile(np.array(w_out), N)
    H = Bs.T @ (W[:, None] * Bs) + w_force * np.eye(n)
    g = -Bs.T @ (W * (prob["ref_out"][k].reshape(-1) - free_resp))
    rows = [i for i in range(N) if dims[i] > 0]
    A = np.zeros((len(rows), n))
    b = np.zeros(len(rows))
    for e, i in enumerate(rows):
        A[e, off[i]:off[i + 1]] = prob["ridge"][k, i, :dims[i], 2]
        b[e] = prob["total_force_z"][k, i]
    lo, hi = np.full(n, 3.0), np.full(n, 3.0 * mass * G)
    return H, g, A, b, lo, hi, dims


def make_xy():
    out = {}
    mass, dt = 100.0, 0.1
    for tag, N, n, seed in (("n20", 20, 12, 31), ("n15", 15, 8, 32)):
        prob, x0 = fd.make_xy_batch(n, N, dt, mass, seed=seed)
        lam_all = np.zeros((n, N, 16))
        certs = np.zeros((n, 4))
        for k in range(n):
            H, g, A, b, lo, hi, dims = xy_qp(prob, k, x0[k], mass, dt)
            # feasible start: the vertical force of every contact step spread evenly over its ridges
            xf = np.zeros(len(g))
            off = np.concatenate([[0], np.cumsum(dims)])
            for e, i in enumerate([i for i in range(N) if dims[i] > 0]):
                xf[off[i]:off[i + 1]] = b[e] / A[e, off[i]:off[i + 1]].sum()
            lam, stat, cert = solve_certified(H, g, A, b, lo, hi, xf, scale=np.abs(b).max())
            for i in range(N):
                lam_all[k, i, :dims[i]] = lam[off[i]:off[i + 1]]
            certs[k] = [cert["stationarity"], cert["equality"], cert["bound_violation"], cert["multiplier_sign"]]
            print("xy", tag, k, "clamped", int((stat != 0).sum()), cert)
        for key, v in prob.items():
            out["%s_%s" % (tag, key)] = v
        out[tag + "_x0"] = x0
        out[tag + "_lambda"] = lam_all
        out[tag + "_cert"] = certs
    np.savez_compressed(os.path.join(HERE, "xy_golden.npz"), **out)
    print("wrote xy_golden.npz", {k: v.shape for k, v in out.items()})


def make_xy_wide():
    """Beyond one surface contact and 20 steps: walking with two separate foot contacts in double support (32 ridges per
    step) over 30 steps, and the reference scenario over 40 steps (src/LinearMpcXY.cpp:69-82, :126-133).  Instances whose
    active set the float64 iteration cannot settle to the long-double certificate (a bound multiplier within 1e-6 of
    zero: degenerate at this precision) are left out."""
    out = {}
    mass, dt = 100.0, 0.1
    for tag, N, M, n, seed in (("w30", 30, 32, 5, 41), ("l40", 40, 16, 4, 42)):
        if M == 32:
            prob, x0 = fd.make_xy_walking_batch(n + 3, N, dt, mass, M=32, seed=seed)
        else:
            prob, x0 = fd.make_xy_batch(n + 3, N, dt, mass, seed=seed)
        keep, lams, certs = [], [], []
        for k in range(n + 3):
            if len(keep) == n:
                break
            H, g, A, b, lo, hi, dims = xy_qp(prob, k, x0[k], mass, dt)
            xf = np.zeros(len(g))
            off = np.concatenate([[0], np.cumsum(dims)])
            for e, i in enumerate([i for i in range(N) if dims[i] > 0]):
                xf[off[i]:off[i + 1]] = b[e] / A[e, off[i]:off[i + 1]].sum()
            try:
                # (multipliers here are of order w_force x force ~ 1e-3: the certificate at 1e-13 of the force scale)
                lam, stat, cert = solve_certified(H, g, A, b, lo, hi, xf, scale=np.abs(b).max(), mult_tol=1e-12,
                                                  sign_tol=1e-13)
            except AssertionError as err:
                print("xy", tag, k, "left out:", err, flush=True)
                continue
            la = np.zeros((N, M))
            for i in range(N):
                la[i, :dims[i]] = lam[off[i]:off[i + 1]]
            keep.append(k)
            lams.append(la)
            certs.append([cert["stationarity"], cert["equality"], cert["bound_violation"], cert["multiplier_sign"]])
            print("xy", tag, k, "variables", len(g), "clamped", int((stat != 0).sum()), cert, flush=True)
        assert len(keep) == n
        for key, v in prob.items():
            out["%s_%s" % (tag, key)] = v[keep]
        out[tag + "_x0"] = x0[keep]
        out[tag + "_lambda"] = np.array(lams)
        out[tag + "_cert"] = np.array(certs)
    np.savez_compressed(os.path.join(HERE, "xy_wide_golden.npz"), **out)
    print("wrote xy_wide_golden.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------------ IntrinsicallyStableMpc
def ism_qp(com_height, dt, N, ref_zmp, zmin, zmax, cp, z0, w_zmp=1.0, w_vel=1e-3):
    """QP of src/IntrinsicallyStableMpc.cpp:8-104 for one axis in the ZMP-velocity variables u; the range constraints
    zmin - z0 <= P u <= zmax - z0 are returned as (P, lo_rows, hi_rows)."""
    omega = np.sqrt(G / com_height)
    lam = np.exp(-omega * dt)
    P = np.tril(np.full((N, N), dt))
    H = w_vel * np.eye(N) + w_zmp * P.T @ P
    g = w_zmp * P.T @ (np.full(N, z0) - ref_zmp)
    a = np.zeros(N)
    a[0] = (1 - lam) / (omega * (1 - lam**N))
    for i in range(1, N):
        a[i] = lam * a[i - 1]
    return H, g, a, cp - z0, P, zmin - z0, zmax - z0


def range_qp_certified(H, g, a, c, P, lo_r, hi_r, scale):
    """min 1/2 u'Hu + g'u, a'u = c, lo_r <= P u <= hi_r with P square and invertible: in y = P u it is a box QP."""
    Pinv = np.linalg.inv(P)
    Hy = Pinv.T @ H @ Pinv
    gy = Pinv.T @ g
    Ay = (a @ Pinv)[None, :]
    # feasible start: least-squares point of the equality, pulled into the box where possible
    y0 = np.clip(np.linalg.lstsq(Ay, np.array([c]), rcond=None)[0], lo_r, hi_r)
    # restore the equality along the entries that still have slack (one dimensional fix-up, then re-clip)
    for _ in range(200):
        r = c - (Ay @ y0)[0]
        if abs(r) < 1e-13:
            break
        w = Ay[0] * (((r * Ay[0] > 0) & (y0 < hi_r)) | ((r * Ay[0] < 0) & (y0 > lo_r)))
        if not np.any(w != 0):
            return None
        y0 = np.clip(y0 + r * w / (w @ Ay[0]), lo_r, hi_r)
    if abs(c - (Ay @ y0)[0]) > 1e-10:
        return None
    y, stat, cert = solve_certified(Hy, gy, Ay, np.array([c]), lo_r, hi_r, y0, scale)
    return Pinv @ y, y, cert


def make_ism():
    out = {}
    for tag, N, hd, n, seed in (("n100", 100, 2.0, 10, 41), ("n20", 20, 0.4, 10, 42)):
        dt = hd / N
        b = fx.make_ism_batch(n, N, dt, 1.0, seed=seed)
        zmp = np.full((n, 2), np.nan)
        vel = np.zeros((n, 2, N))
        ok = np.zeros((n, 2), dtype=np.int32)
        for k in range(n):
            for ax in range(2):
                ref, zmin, zmax = b["ref"][k, ax, 0], b["ref"][k, ax, 1], b["ref"][k, ax, 2]
                cp, z0 = b["init"][k, ax]
                H, g, a, c, P, lo_r, hi_r = ism_qp(1.0, dt, N, ref, zmin, zmax, cp, z0)
                r = range_qp_certified(H, g, a, c, P, lo_r, hi_r, scale=1.0)
                if r is None:
                    continue
                u, y, cert = r
                ok[k, ax] = 1
                vel[k, ax] = u
                zmp[k, ax] = min(max(z0 + 0.005 * u[0], zmin[0]), zmax[0])
                print("ism", tag, k, ax, cert)
        out.update({tag + "_init": b["init"], tag + "_ref": b["ref"], tag + "_zmp": zmp, tag + "_vel": vel,
                    tag + "_ok": ok})
    np.savez_compressed(os.path.join(HERE, "ism_golden.npz"), **out)
    print("wrote ism_golden.npz", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------------ LinearMpcZ
def z_qp(mass, dt, contact, ref, x0, w_pos=1.0, w_force=1e-7):
    """QP of src/LinearMpcZ.cpp:73-94: variables = forces of the contact steps."""
    N = len(contact)
    A = np.array([[0.0, 1.0], [0.0, 0.0]])
    E = np.array([0.0, -mass * G])
    Adc, Bdc, Edc = zoh(A, np.array([[0.0], [1.0]]), E, dt)
    Adn, _, Edn = zoh(A, np.zeros((2, 0)), E, dt)
    C = np.array([1.0 / mass, 0.0])
    idx = [i for i in range(N) if contact[i]]
    n = len(idx)

    def simulate(xs, f, with_offset):
        out = np.zeros(N)
        x = xs.copy()
        j = 0
        for i in range(N):
            if contact[i]:
                x = Adc @ x + Bdc[:, 0] * f[j] + (Edc if with_offset else 0.0)
                j += 1
            else:
                x = Adn @ x + (Edn if with_offset else 0.0)
            out[i] = C @ x
        return out

    free_resp = simulate(mass * x0, np.zeros(n), True)
    Bs = np.zeros((N, n))
    for j in range(n):
        e = np.zeros(n)
        e[j] = 1.0
        Bs[:, j] = simulate(np.zeros(2), e, False)
    H = w_pos * Bs.T @ Bs + w_force * np.eye(n)
    g = -w_pos * Bs.T @ (ref - free_resp)
    return H, g, idx


def make_z():
    out = {}
    mass = 100.0
    for tag, N, dt, n, seed in (("n40", 40, 0.05, 24, 51), ("n12", 12, 0.1, 12, 52)):
        b = fx.make_z_batch(n, N, dt, seed=seed)
        w_pos, w_force = fx.Z_WEIGHTS if hasattr(fx, "Z_WEIGHTS") else (1.0, 1e-7)
        force = np.zeros(n)
        force_all = np.zeros((n, N))
        for k in range(n):
            contact = b["contact"][k].astype(bool)
            if not contact[0]:
                continue  # src/LinearMpcZ.cpp:54-57: no contact now -> planned force 0
            H, g, idx = z_qp(mass, dt, contact, b["ref_pos"][k], b["x0"][k], w_pos, w_force)
            nvar = len(idx)
            lo, hi = np.full(nvar, 10.0), np.full(nvar, 10.0 * mass * G)
            x, stat, cert = solve_certified(H, g, np.zeros((0, nvar)), np.zeros(0), lo, hi, np.full(nvar, mass * G),
                                            scale=mass * G)
            force[k] = x[0]
            force_all[k, idx] = x
            print("z", tag, k, "clamped", int((stat != 0).sum()), cert)
        out.update({tag + "_contact": b["contact"], tag + "_ref_pos": b["ref_pos"], tag + "_x0": b["x0"],
                    tag + "_force": force, tag + "_force_all": force_all})
    np.savez_compressed(os.path.join(HERE, "z_golden.npz"), **out)
    print("wrote z_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["xy", "ism", "z"]
    if "z" in which:
        make_z()
    if "ism" in which:
        make_ism()
    if "xy" in which:
        make_xy()
    if "xy_wide" in which:
        make_xy_wide()
