"""Time LinearMpcZmp at the reference test's horizon (2 s @ 20 ms = 100 steps: the workgroup-per-QP kernel).
usage: zmp_n100_bench.py [n] [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base = min(n, 2048)
b = fx.make_zmp_batch(base, 100, 0.02, seed=5)
k = (n + base - 1) // base
x0 = torch.from_numpy(np.concatenate([b["x0"]] * k)[:n]).to("cuda:0")
zl = torch.from_numpy(np.concatenate([b["zlim"]] * k)[:n]).to("cuda:0")
mpc = LinearMpcZmp(1.0, 2.0, 0.02)
z = torch.zeros((n, 2), dtype=torch.float64, device="cuda:0")
st = torch.zeros((n, 2), dtype=torch.int32, device="cuda:0")
mpc.plan_batch_device(x0, zl, 0.005, z, None, st)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); mpc.plan_batch_device(x0, zl, 0.005, z, None, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts); s = st.cpu().numpy()
print("LinearMpcZmp n=%d N=100: %.1f ms -> %.0f solves/s (mean pivots/axis %.1f, non-ok %d)" % (n, t * 1e3, n / t, (s >> 8).mean(), int(((s & 0xff) != 0).sum())))
