"""How the block iteration of the LinearMpcXY stage-recursion kernel ends, per instance, under a given iteration cap
(CCC_XY_PDAS_ITERS / CCC_XY_ROUNDS): converged after k changes of the clamped set, or handed to the dual kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd
cap = int(os.environ.get("CCC_XY_PDAS_ITERS", "16"))
n, N, dt = 65536, 20, 0.1
prob, x0 = fd.make_xy_batch(n, N, dt, seed=20250928)
mpc = LinearMpcXY(100.0, dt, N)
dev = torch.device("cuda:0")
tp = {a: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for a, v in prob.items()}
u0 = torch.zeros((n, 16), dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev)
mpc.plan_batch_device(tp, torch.from_numpy(x0).to(dev), u0, status=st)
torch.cuda.synchronize()
p = (st.cpu().numpy() >> 8)
print("cap %d: converged after k changes:" % cap, np.bincount(p[p <= cap], minlength=cap + 1).tolist())
q = p[p > cap]
print("handed to the dual kernel: %d instances, pivots min/median/max = %s" % (q.size, (q.min(), np.median(q), q.max()) if q.size else "-"))
