"""Time the IntrinsicallyStableMpc kernel (reference test horizon: 2 s @ 20 ms = 100 steps).
The CPU baseline and the parity check live in `python bench.py --workload ism`.  usage: ism_bench.py [n] [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import IntrinsicallyStableMpc, fixtures as fx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base = min(n, 1024)
b = fx.make_ism_batch(base, 100, 0.02, seed=7)
k = (n + base - 1) // base
init = np.concatenate([b["init"]] * k)[:n]
ref = np.concatenate([b["ref"]] * k)[:n]
mpc = IntrinsicallyStableMpc(1.0, 2.0, 0.02)
dev = torch.device("cuda:0")
ti, tr = torch.from_numpy(init).to(dev), torch.from_numpy(ref).to(dev)
z = torch.zeros((n, 2), dtype=torch.float64, device=dev)
st = torch.zeros((n, 2), dtype=torch.int32, device=dev)
mpc.plan_batch_device(ti, tr, 0.005, z, status=st)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); mpc.plan_batch_device(ti, tr, 0.005, z, status=st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
s = st.cpu().numpy()
print("IntrinsicallyStableMpc n=%d N=100: %.3f ms -> %.0f solves/s (mean pivots/axis %.1f, max %d, non-ok %d)"
      % (n, t * 1e3, n / t, (s >> 8).mean(), (s >> 8).max(), int(((s & 0xff) != 0).sum())))
