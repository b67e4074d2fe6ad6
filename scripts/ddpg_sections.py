"""Per-section cycle breakdown of the DDP GROUP kernel (csrc/ddp_group.h).  Needs a profiling build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCCC_DDPG_PROF centroidalcontrolcollection_amd/csrc/*.hip -o scratch/libccc_ddpg_prof.so
    CCC_AMD_LIB=$PWD/scratch/libccc_ddpg_prof.so python scripts/ddpg_sections.py [n] [cen|srb]
(the profiling build returns the section timings in place of the first planned inputs; cycles are those of a WAVEFRONT,
i.e. of four instances in lock-step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
srb = len(sys.argv) > 2 and sys.argv[2] == "srb"
N, dt = (50, 0.03) if srb else (100, 0.03)
prob, x0 = fd.make_centroidal_batch(n, N, dt, seed=1, srb=srb)
if srb:
    d = DdpSingleRigidBody(100.0, dt, N, DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                                                         terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3))
else:
    d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
d.ddp_solver_.config().max_iter = 20
r = d.planOnceBatch(prob, x0)
tm = r["u"][:, 0, :16].mean(axis=0)
names = ["rollouts", "backward passes (total)", "  derivatives + products", "  box-QP", "    of which Cholesky", "  gains",
         "  value update"]
tot = tm[7]
for k, nm in enumerate(names):
    print("%-28s %14.0f cycles  %5.1f %%" % (nm, tm[k], 100 * tm[k] / tot))
print("TOTAL %.0f cycles per wavefront (4 instances) = %.1f ms at 2.4 GHz" % (tot, tot / 2.4e6))
print("rollouts %.1f, backward passes %.1f, box-QP iterations %.0f (%.2f per backward step), factorisations %.0f (%.2f), "
      "line-search steps %.0f" % (tm[8], tm[9], tm[10], tm[10] / (tm[9] * N), tm[11], tm[11] / (tm[9] * N), tm[12]))
print("per backward step: %.0f cycles; per rollout step: %.0f cycles" % (tm[1] / (tm[9] * N), tm[0] / (tm[8] * N)))
