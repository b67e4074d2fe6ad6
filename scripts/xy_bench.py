"""Time the LinearMpcXY kernel on BASELINE config 4 (N=20, dt=0.1, batch 65536 by default; pass a smaller n to probe)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, dt = 20, 0.1
base = min(n, 2048)
prob, x0 = fd.make_xy_batch(base, N, dt, seed=7)
k = (n + base - 1) // base
prob = {a: np.concatenate([v] * k)[:n] for a, v in prob.items()}
x0 = np.concatenate([x0] * k)[:n]
mpc = LinearMpcXY(100.0, dt, N)
dev = torch.device("cuda:0")
tp = {a: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for a, v in prob.items()}
tx0 = torch.from_numpy(x0).to(dev)
u0 = torch.zeros((n, 16), dtype=torch.float64, device=dev)
st = torch.zeros(n, dtype=torch.int32, device=dev)
mpc.plan_batch_device(tp, tx0, u0, status=st)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); mpc.plan_batch_device(tp, tx0, u0, status=st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
s = st.cpu().numpy()
print("LinearMpcXY n=%d N=%d: %.1f ms -> %.0f solves/s (mean pivots %.1f, max %d, non-ok %d)"
      % (n, N, t * 1e3, n / t, (s >> 8).mean(), (s >> 8).max(), int(((s & 0xff) != 0).sum())))
