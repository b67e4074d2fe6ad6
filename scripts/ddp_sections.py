"""Per-section cycle breakdown of the DDP kernel.  Needs a profiling build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCCC_DDP_PROF centroidalcontrolcollection_amd/csrc/*.hip -o scratch/libccc_ddp_prof.so
    CCC_AMD_LIB=$PWD/scratch/libccc_ddp_prof.so python scripts/ddp_sections.py [n] [cen|srb|walk]
(the profiling build returns the section timings in place of the first planned inputs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
srb = len(sys.argv) > 2 and sys.argv[2] == "srb"
walk = len(sys.argv) > 2 and sys.argv[2] == "walk"  # double-support walking (32 ridges per step)
N, dt = (50, 0.03) if srb else (100, 0.03)
kw = {}
if walk:
    N, dt = 40, 0.05
    prob, x0 = fd.make_walking_batch(n, N, dt, seed=1)
    kw = dict(max_phases=prob["phase_dim"].shape[1], max_ridges=32)
else:
    prob, x0 = fd.make_centroidal_batch(n, N, dt, seed=1, srb=srb)
if srb:
    d = DdpSingleRigidBody(100.0, dt, N, DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                                                         terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3))
else:
    d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)), **kw)
d.ddp_solver_.config().max_iter = 20
r = d.planOnceBatch(prob, x0)
tm = r["u"][:, 0, :16].mean(axis=0)
names = ["deriv", "products", "boxqp value_of", "boxqp gradient/flags", "boxqp cholesky", "boxqp solve", "boxqp line search",
         "gains", "value update", "rollouts (line search)", "other"]
tot = tm[15]
for k, nm in enumerate(names):
    print("%-26s %12.0f cycles  %5.1f %%" % (nm, tm[k], 100 * tm[k] / tot))
print("%-26s %12.0f cycles  %5.1f %%" % ("(unattributed)", tot - tm[:11].sum(), 100 * (tot - tm[:11].sum()) / tot))
print("TOTAL %.0f cycles per instance; iterations %.2f" % (tot, r["iters"].mean()))
steps = r["iters"].mean() * N
print("box-QP: %.2f calls per backward step, %.2f iterations per call, %.2f factorisations per call; %.2f rollouts per iteration"
      % (tm[11] / steps, tm[12] / tm[11], tm[13] / tm[11], tm[14] / r["iters"].mean()))
