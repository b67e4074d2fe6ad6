"""Per-section cycle breakdown of the tile DDP kernel (csrc/ddp_tile.h).  Needs a profiling build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCCC_TILE_PROF centroidalcontrolcollection_amd/csrc/*.hip -o scratch/libccc_tile_prof.so
    CCC_AMD_LIB=$PWD/scratch/libccc_tile_prof.so python scripts/ddp_tile_sections.py [n] [cen|srb]
(the profiling build returns the section timings in place of the first planned inputs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
srb = len(sys.argv) > 2 and sys.argv[2] == "srb"
N, dt = (50, 0.03) if srb else (100, 0.03)
prob, x0 = fd.make_centroidal_batch(min(n, 2048), N, dt, seed=1, srb=srb)
k = (n + 2047) // 2048
prob = {a: np.concatenate([v] * k)[:n] for a, v in prob.items()}
x0 = np.concatenate([x0] * k)[:n]
if srb:
    d = DdpSingleRigidBody(100.0, dt, N, DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                                                         terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3))
else:
    d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
d.ddp_solver_.config().max_iter = 20
r = d.planOnceBatch(prob, x0)
tm = np.concatenate([r["u"][:, 0, :16], r["u"][:, 1, :16]], axis=1).mean(axis=0)
names = ["derivatives", "products", "boxqp value_of", "boxqp gradient/flags", "boxqp factorisation", "boxqp solve",
         "boxqp line search", "gains", "value update", "forward passes", "boxqp entry/exit"]
tm[10] -= tm[2:7].sum()  # (the box-QP slice of backward_step() spans its inner sections)
tot = tm[:11].sum()
for j, nm in enumerate(names):
    print("%-26s %12.0f cycles  %5.1f %%" % (nm, tm[j], 100 * tm[j] / tot))
steps = r["iters"].mean() * N
print("TOTAL (attributed) %.0f cycles per instance; iterations %.2f; %.0f cycles per backward step" % (tot, r["iters"].mean(), (tot - tm[9]) / steps))
print("box-QP: %.2f calls per backward step, %.2f iterations per call, %.2f factorisations per call; %.2f forward passes per iteration, %.0f cycles per forward step"
      % (tm[11] / steps, tm[12] / tm[11], tm[13] / tm[11], tm[14] / r["iters"].mean(), tm[9] / (tm[14] * N)))

print("factorisation split: C_f %.0f, M_f %.0f, Gauss-Jordan %.0f cycles per factorisation" % tuple(tm[15 + q] / tm[13] for q in range(3)))
