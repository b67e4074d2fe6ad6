"""Monte-Carlo closed loop of TestLinearMpcXY.cpp:98-132 on the device: n instances of the reference's contact sequence from
perturbed initial states (position +- 5 cm, so that the problems differ from instance to instance and move from cycle to
cycle), `cycles` control cycles in one call (ccc_xy_closed_loop_device).  usage: xy_loop_bench.py [n] [cycles]
(CCC_XY_HISTORY=0 in the environment: every cycle in the caller's order)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcXY, centroidal_loop as cl, fixtures_ddp as fd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N, dt, mass = 20, 0.1, 100.0
dev = torch.device("cuda:0")
mpc = LinearMpcXY(mass, dt, N)
rects = [((0.9, -0.15), (1.1, 0.15)), ((0.9, 0.05), (1.1, 0.15)), ((1.15, -0.15), (1.35, -0.05)),
         ((1.4, 0.05), (1.6, 0.15)), ((1.4, -0.15), (1.6, 0.15))]
refs = [(1.0, 0.0), (1.0, 0.1), (1.25, -0.1), (1.5, 0.1), (1.5, 0.0)]
K = C = 5
seg_end = np.tile(np.array([3.0, 4.0, 5.0, 6.0, 1e30]), (n, 1))
seg_contact = np.tile(np.arange(5, dtype=np.int32), (n, 1))
seg_ref = np.zeros((n, K, 6))
dim = np.full((n, C), 16, dtype=np.int32)
vert, ridge = np.zeros((n, C, 16, 3)), np.zeros((n, C, 16, 3))
for c in range(5):
    vert[:, c], ridge[:, c] = fd.contact_from_rect(*rects[c])
    seg_ref[:, c, :3] = [refs[c][0], refs[c][1], 1.0]
tl = cl.ContactTimeline(seg_end, seg_contact, seg_ref, dim, vert, ridge, 0.0)
state0 = np.zeros((n, 18)); state0[:, :3] = (1.0, 0.0, 1.0)
state0[:, :2] += 0.05 * np.random.default_rng(4).uniform(-1, 1, size=(n, 2))
inertia = torch.from_numpy(np.tile(np.array((40.0, 20.0, 10.0)), (n, 1))).to(dev)
for rep in range(2):
    sim = torch.from_numpy(state0.copy()).to(dev)
    stats = torch.zeros((n, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t_end = cl.xy_closed_loop(mpc, tl, 1.0, inertia, sim, 2.0, 0.05, cycles, stats=stats)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
fin = sim.cpu().numpy()
print("XY closed loop n=%d cycles=%d (t = 2.0 .. %.2f s): %.1f ms -> %.2f M instance-cycles/s (%.2f ms per cycle); final |pos - mean| max %.3f m, checksum %.9f"
      % (n, cycles, t_end, t * 1e3, n * cycles / t / 1e6, t * 1e3 / cycles, np.abs(fin[:, :3] - fin[:, :3].mean(axis=0)).max(), float(fin[:, :3].sum())))
