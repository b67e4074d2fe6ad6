"""Development aid: per-instance start time and duration of the DDP kernel (library built with -DCCC_TILE_TIMING:
scripts/unit_variant.sh ddp_tile timing -DCCC_TILE_TIMING), to see what a batch's makespan is made of.
usage: CCC_AMD_LIB=scratch/libccc_timing.so python scripts/ddp_sched_probe.py [n] [cen|srb]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import DdpCentroidal, DdpSingleRigidBody, fixtures_ddp as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
srb = (sys.argv[2] if len(sys.argv) > 2 else "srb") == "srb"
N, dt = (50, 0.03) if srb else (100, 0.03)
base = min(n, 4096)
prob, x0 = fd.make_centroidal_batch(base, N, dt, seed=1, srb=srb)
k = (n + base - 1) // base
prob = {a: np.concatenate([v] * k)[:n] for a, v in prob.items()}
x0 = np.concatenate([x0] * k)[:n]
if srb:
    d = DdpSingleRigidBody(100.0, dt, N, DdpSingleRigidBody.WeightParam(running_pos=(1.0, 1.0, 10.0), running_ori=(0.5,) * 3,
                                                                         terminal_pos=(1.0, 1.0, 10.0), terminal_ori=(0.5,) * 3))
else:
    d = DdpCentroidal(100.0, dt, N, DdpCentroidal.WeightParam(running_pos=(1, 1, 10), terminal_pos=(1, 1, 10)))
d.ddp_solver_.config().max_iter = 20
dev = torch.device("cuda:0")
tp = {a: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for a, v in prob.items()}
tx0 = torch.from_numpy(x0).to(dev)
u = torch.zeros((n, N, 16), dtype=torch.float64, device=dev)
it = torch.zeros(n, dtype=torch.int32, device=dev)
cost = torch.zeros(n, dtype=torch.float64, device=dev)
for _ in range(2):
    d.plan_batch_device(tp, tx0, u, iters=it, cost=cost)
    torch.cuda.synchronize()
c = cost.cpu().numpy(); its = it.cpu().numpy(); uu = u.cpu().numpy()
dur = c; start = uu[:, 0, 0]; end = uu[:, 0, 1]
t00 = start.min(); start = start - t00; end = end - t00
ms = lambda t: t / 1e5   # 100 MHz ticks
print("n=%d: makespan %.1f ms; busy per instance mean %.2f median %.2f p90 %.2f p99 %.2f max %.2f ms; busy sum / 2048 slots = %.1f ms"
      % (n, ms(end.max()), ms(dur.mean()), ms(np.median(dur)), ms(np.percentile(dur, 90)), ms(np.percentile(dur, 99)), ms(dur.max()), ms(dur.sum()) / 2048))
fin = np.sort(ms(end))
print("instances unfinished at 50/60/70/80/90/95/100 %% of the makespan:", [int((fin > q * fin[-1]).sum()) for q in (0.5, 0.6, 0.7, 0.8, 0.9, 0.95, 0.999)])
print("last to finish (iters, busy ms, first start ms, end ms):", [(int(its[i]), round(ms(dur[i]), 1), round(ms(start[i]), 1), round(ms(end[i]), 1)) for i in np.argsort(-end)[:8]])
