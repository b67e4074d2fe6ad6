"""Run the LinearMpcXY kernel on a seeded batch and save all force scales (offline accuracy analysis)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from centroidalcontrolcollection_amd import LinearMpcXY, fixtures_ddp as fd
N = int(sys.argv[1]); n = int(sys.argv[2]); seed = int(sys.argv[3])
prob, x0 = fd.make_xy_batch(n, N, 0.1, seed=seed)
r = LinearMpcXY(100.0, 0.1, N).planOnceBatch(prob, x0, want_all=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/xy_dump_%d_%d_%d.npz" % (N, n, seed), lam=r["lam"], status=r["status"], pivots=r["pivots"])
