"""Development aid: what makes the slowest instances of a DDP batch slow -- per-instance counters of the profiling build
(scripts/unit_variant.sh ddp_tile prof -DCCC_TILE_PROF).  usage: CCC_AMD_LIB=scratch/libccc_prof.so CCC_DDP_SLICE=0 python scripts/ddp_slow_instances.py [cen|srb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd
srb = len(sys.argv) > 1 and sys.argv[1] == "srb"
n = 2048
N, dt = (50, 0.03) if srb else (100, 0.03)
prob, x0 = fd.make_centroidal_batch(n, N, dt, seed=1, srb=srb)
if srb:
    d = DdpSingleRigidBody(100.0, dt, N, DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                                                         terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3))
else:
    d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
d.ddp_solver_.config().max_iter = 20
r = d.planOnceBatch(prob, x0)
tm = np.concatenate([r["u"][:, 0, :16], r["u"][:, 1, :16]], axis=1)
tot = tm[:, :11].sum(axis=1) - tm[:, 2:7].sum(axis=1)   # (entry 10 spans the box-QP's inner sections)
its = r["iters"]
order = np.argsort(-tot)
def row(i):
    return ("iters %2d status %2d | Mcycles %6.1f | qp calls %5d iters %6d factors %5d | forward passes %3d | per-iteration: qp iters %.0f, forwards %.1f"
            % (its[i], r["status"][i], tot[i] / 1e6, tm[i, 11], tm[i, 12], tm[i, 13], tm[i, 14], tm[i, 12] / max(its[i], 1), tm[i, 14] / max(its[i], 1)))
print("slowest:")
for i in order[:6]: print("  ", row(i))
print("median:")
for i in order[n // 2 - 2:n // 2 + 2]: print("  ", row(i))
print("fastest of the 20-iteration ones:")
full = [i for i in order[::-1] if its[i] == 20][:3]
for i in full: print("  ", row(i))
bs = N * its
print("correlation of cycles with: qp iters %.3f, forwards %.3f, iters %.3f" % (np.corrcoef(tot, tm[:, 12])[0, 1], np.corrcoef(tot, tm[:, 14])[0, 1], np.corrcoef(tot, its)[0, 1]))
names = ["derivatives", "products", "qp value_of", "qp gradient", "qp factor", "qp solve", "qp search", "gains", "value update", "forward", "qp entry/exit"]
for lab, idx in (("slowest 20", order[:20]), ("middle 200", order[n // 2 - 100:n // 2 + 100])):
    m = tm[idx].mean(axis=0); m[10] -= m[2:7].sum()
    fresh = m[13] - (m[13] - m[11])  # (first-iteration factorisations = box-QP calls)
    print(lab, "factor detail: Cf %.0f Mf %.0f GJ %.0f cycles per fresh factorisation (sum over the instance %.1f M); rank-one updates: %.0f per instance, "
          "%.0f cycles each (%.1f M)" % (m[15] / max(m[13], 1), m[16] / max(m[13], 1), m[17] / max(m[13], 1), (m[15] + m[16] + m[17]) / 1e6,
                                            m[19], m[18] / max(m[19], 1), m[18] / 1e6))
    print(lab, " ".join("%s %.1f%%" % (nm, 100 * m[j] / m[:11].sum()) for j, nm in enumerate(names)))
    steps = max(m[14], 1) * N  # forward passes x steps
    print(lab, "forward step, cycles: fetch issue %.0f, feedback + store %.0f, running cost %.0f, terms %.0f, state + store %.0f (sum %.0f; forward total / steps %.0f)"
          % (m[20] / steps, m[21] / steps, m[22] / steps, m[23] / steps, m[24] / steps, m[20:25].sum() / steps, m[9] / steps))
