"""Monte-Carlo closed loop on the device: n random footstep timelines x `cycles` control cycles of
TestLinearMpcZmp.cpp:55-102 (sample limits -> planOnce -> simulate -> disturb), one call.  usage: zmp_loop_bench.py [n] [cycles]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tl = {k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in fx.make_zmp_timelines(n, seed=1).items()}
mpc = LinearMpcZmp(1.0, 2.0, float(os.environ.get("CCC_LOOP_DT", "0.0625")))  # (CCC_LOOP_DT=0.02: the reference test's own 100-step horizon)
rng = np.random.default_rng(0)
com0 = np.zeros((n, 2, 2)); com0[:, :, 0] = rng.uniform(-0.02, 0.02, size=(n, 2)); com0[:, :, 1] = rng.uniform(-0.05, 0.05, size=(n, 2))
for rep in range(2):
    com = torch.from_numpy(com0.copy()).to("cuda:0"); zmp = torch.from_numpy(com0[:, :, 0].copy()).to("cuda:0")
    viol = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mpc.closed_loop_device(tl, com, zmp, 0.0, 0.02, cycles, disturb_times=(2.0, 5.0), disturb_impulse=0.05, violations=viol)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
v = viol.cpu().numpy()
print("closed loop n=%d cycles=%d: %.1f ms -> %.1f M instance-cycles/s (%.3f ms per cycle); instances with a limit violation: %d; "
      "CoM still bounded (|pos| < 5 m): %d" % (n, cycles, t * 1e3, n * cycles / t / 1e6, t * 1e3 / cycles, int((v > 0).sum()),
                                                int((com.abs().amax(dim=(1, 2)) < 5).sum())))
