"""Run on an MI355X: the pivot trips per QP the headline kernel spends on the bench workload (seeds 20250928, 20250929) ->
tests/tools/zmp_pivot_counts.npz, the data tests/tools/zmp_trip_predictor.py and zmp_gi_model.py work on.
"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
mpc = LinearMpcZmp(1.0, 2.0, 0.0625)
dev = torch.device("cuda:0")
out = {}
for k in range(2):
    b = fx.make_zmp_batch(65536, 32, 0.0625, 1.0, seed=20250928 + k)
    x0 = torch.from_numpy(b["x0"]).to(dev); zl = torch.from_numpy(b["zlim"]).to(dev)
    z = torch.empty((65536, 2), dtype=torch.float64, device=dev); st = torch.empty((65536, 2), dtype=torch.int32, device=dev)
    mpc.plan_batch_device(x0, zl, 0.005, z, None, st, torch.cuda.current_stream())
    torch.cuda.synchronize()
    out["piv%d" % k] = (st.cpu().numpy() >> 8).astype(np.int16)
np.savez_compressed("gpurun_out/zmp_pivot_counts.npz", **out)
print({k: (v.mean(), v.max()) for k, v in out.items()})
