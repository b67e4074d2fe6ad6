"""Time the DdpZmp closed loop on the device (plan -> simulate -> plan ..., one launch).  usage: ddpzmp_loop_bench.py [n] [cycles]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import DdpZmp, fixtures as fx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N, dt, mass, h, sim_dt = 100, 0.02, 100.0, 1.0, 0.005
fm = fx.FootstepManager()
for fs in fx.reference_scenario_footsteps():
    fm.appendFootstep(fs)
fm.update(0.0)
kt, kz = np.array(fm._zmp_times), np.array(fm._zmps)
rng = np.random.default_rng(1)
state = np.zeros((6, n)); state[4] = h
state[0] = rng.uniform(-0.01, 0.01, n); state[2] = rng.uniform(-0.01, 0.01, n)
dev = torch.device("cuda:0")
d = DdpZmp(mass, dt, N); d.ddp_solver_.config().max_iter = 3
tk = torch.from_numpy(np.repeat(kt[:, None], n, axis=1).copy()).to(dev)
tz = torch.from_numpy(np.repeat(kz[:, :, None], n, axis=2).copy()).to(dev)
stats = torch.zeros((4, n), dtype=torch.float64, device=dev)
t0 = 1.9  # just before the first footstep: the planner has work to do
for rep in range(2):
    ts = torch.from_numpy(state.copy()).to(dev)
    torch.cuda.synchronize(); a = time.perf_counter()
    d.closed_loop_device(tk, tz, h, ts, t0, sim_dt, cycles, (), 0.0, stats, None)
    torch.cuda.synchronize(); t = time.perf_counter() - a
st = stats.cpu().numpy()
print("DdpZmp closed loop n=%d cycles=%d: %.1f ms -> %.2f M instance-cycles/s (%.2f DDP iterations per cycle; worst |zmp - ref| %.3f m)"
      % (n, cycles, t * 1e3, n * cycles / t / 1e6, st[3].mean() / cycles, st[0].max()))
