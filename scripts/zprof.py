"""Per-phase cycle counts and per-QP wall-clock spans of the LDS-tableau LinearMpcZmp kernels; needs the library built
with -DCCC_ZMP_PROF (see csrc/zmp.hip).  usage: zprof.py N [grid]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
N = int(sys.argv[1]); n = 8192; dt = 2.0 / N
b = fx.make_zmp_batch(2048, N, dt, seed=5)
x0 = torch.from_numpy(np.concatenate([b["x0"]] * 4)).to("cuda:0")
zl = torch.from_numpy(np.concatenate([b["zlim"]] * 4)).to("cuda:0")
mpc = LinearMpcZmp(1.0, 2.0, dt)
z = torch.zeros((n, 2), dtype=torch.float64, device="cuda:0")
st = torch.zeros((n, 2), dtype=torch.int32, device="cuda:0")
jerk = torch.zeros((n, 2, N), dtype=torch.float64, device="cuda:0")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mpc.plan_batch_device(x0, zl, 0.005, z, jerk, st)
    torch.cuda.synchronize(); print("call %.2f ms" % ((time.perf_counter() - t0) * 1e3))
j = jerk.reshape(-1, N).cpu().numpy(); piv = (st.reshape(-1).cpu().numpy() >> 8)
rt = j[:, 6] / 100.0; start = j[:, 8] / 100.0; start -= start.min()
print("per-QP us: mean %.1f max %.1f ; pivots mean %.1f ; us/pivot %.2f" % (rt.mean(), rt.max(), piv.mean(), rt.sum() / piv.sum()))
print("sections per pivot (cycles):", np.round(j[:, :6].sum(0) / piv.sum()))
print("kernel span from first start to last end: %.1f us; sum(QP time)/span = %.1f concurrent WGs" % ((start + rt).max(), rt.sum() / (start + rt).max()))
setup = j[:, 9] / 100.0
nxt = {}
print("setup us mean %.1f" % setup.mean())
# per block: QPs are qp = blk + k*grid ; estimate the gap between consecutive QPs of a block = output phase of previous
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if grid:
    s0 = j[:, 8] / 100.0
    gaps = []
    for blk in range(0, grid, 37):
        idx = np.arange(blk, j.shape[0], grid)
        for a_, b_ in zip(idx[:-1], idx[1:]):
            gaps.append(s0[b_] - (s0[a_] + setup[a_] + rt[a_]))
    print("output-phase us (gap to next QP start) mean %.1f" % np.mean(gaps))
if grid:
    bs = np.sort(s0[:grid] - s0[:grid].min())
    print("block start times us: ", np.round(bs[[0, 100, 200, 255, 256, 300, 400, 511, 512, 600, 800, 1023, 1024, 1500, 2047]], 0))
    be = (s0 + setup + rt)[-grid:] - s0.min()
    print("block end times us (last QP of each block) percentiles:", np.round(np.percentile(be, [0, 10, 50, 90, 100]), 0))
if grid:
    t0_ = s0.min()
    bstart = s0[:grid] - t0_
    nq = j.shape[0]
    last = np.array([np.arange(b_, nq, grid)[-1] for b_ in range(grid)])
    bend = (s0 + setup + rt)[last] - t0_
    span = bend.max()
    for f in (0.02, 0.1, 0.25, 0.5, 0.75, 0.9, 0.98):
        t = f * span
        print("t=%.0f us: %d blocks resident" % (t, int(((bstart <= t) & (bend > t)).sum())))
    print("block duration us: mean %.0f  min %.0f max %.0f" % ((bend - bstart).mean(), (bend - bstart).min(), (bend - bstart).max()))
