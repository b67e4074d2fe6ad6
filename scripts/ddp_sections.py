"""Per-section cycle breakdown of the DDP kernel (needs the instrumented build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import DdpCentroidal, fixtures_ddp as fd
n, N, dt = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 100, 0.03
prob, x0 = fd.make_centroidal_batch(n, N, dt, seed=1)
d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
d.ddp_solver_.config().max_iter = 20
r = d.planOnceBatch(prob, x0)
tm = r["u"][:, 0, :12].mean(axis=0)
names = ["deriv", "products", "boxqp(total)", "gains", "value update", "  chol", "  solve", "rollouts", "TOTAL"]
for k, nm in enumerate(names):
    print("%-14s %12.0f cycles  %5.1f %%" % (nm, tm[k], 100 * tm[k] / tm[8]))
print("iters mean", r["iters"].mean())
