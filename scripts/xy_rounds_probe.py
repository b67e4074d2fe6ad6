"""Histogram of the stage-recursion kernel's iteration counts (LinearMpcXY, N=20) -- used to choose the rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd
n, N, dt = 65536, 20, 0.1
prob, x0 = fd.make_xy_batch(2048, N, dt, seed=7)
k = n // 2048
prob = {a: np.concatenate([v] * k) for a, v in prob.items()}
x0 = np.concatenate([x0] * k)
mpc = LinearMpcXY(100.0, dt, N)
dev = torch.device("cuda:0")
tp = {a: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for a, v in prob.items()}
u0 = torch.zeros((n, 16), dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev)
mpc.plan_batch_device(tp, torch.from_numpy(x0).to(dev), u0, status=st)
torch.cuda.synchronize()
p = (st.cpu().numpy() >> 8)
print("changes of the clamped set:", np.bincount(np.minimum(p, 20)).tolist())
q = p[p >= 17]
print("handed to the dual kernel: %d instances (%d distinct of the 2048 tiled), pivots min/median/p90/max = %d/%d/%d/%d"
      % (q.size, np.unique(np.nonzero(p >= 17)[0] % 2048).size, q.min(), np.median(q), np.percentile(q, 90), q.max()))
print("pivot histogram (bins of 20):", np.bincount(q // 20).tolist())
