"""Time the DdpZmp kernel (one instance per lane).  usage: ddpzmp_bench.py [n] [max_iter] [N] [reps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import DdpZmp, fixtures as fx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dt = 0.02
base = min(n, 2048)
b = fx.make_ddpzmp_batch(base, N, dt, seed=1)
k = (n + base - 1) // base
dev = torch.device("cuda:0")
ref = torch.from_numpy(np.concatenate([b["ref"]] * k)[:n]).to(dev)
x0 = torch.from_numpy(np.concatenate([b["x0"]] * k)[:n]).to(dev)
ui = torch.from_numpy(np.concatenate([b["u_init"]] * k)[:n]).to(dev)
d = DdpZmp(100.0, dt, N)
d.ddp_solver_.config().max_iter = max_iter
u = torch.zeros((n, N, 3), dtype=torch.float64, device=dev)
it = torch.zeros(n, dtype=torch.int32, device=dev)
d.plan_batch_device(ref, x0, ui, u, iters=it)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); d.plan_batch_device(ref, x0, ui, u, iters=it); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
iters = it.float().mean().item()
# bytes a lane moves: set-up (ref + u_init in, rollout out), per iteration one backward (13 in, 21 out) and >= 1 forward
# (34 in, 9 out) per step, outputs
bytes_ = n * 8.0 * ((N + 1) * 4 * 2 + N * 3 * 2 + (N + 1) * 6 + iters * N * (13 + 21 + 34 + 9) + N * 3 * 2)
print("DdpZmp n=%d N=%d max_iter=%d: %.2f ms -> %.0f solves/s (mean iters %.2f); >= %.1f GB moved -> %.0f GB/s; workspace %.2f GB"
      % (n, N, max_iter, t * 1e3, n / t, iters, bytes_ / 1e9, bytes_ / t / 1e9, d.workspace_bytes(n) / 1e9))
