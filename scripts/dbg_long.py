import numpy as np, os, sys
sys.path.insert(0,'.')
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
from oracle import oracle
b = fx.make_zmp_batch(400, 100, 0.05, seed=5)
ref = oracle.LinearMpcZmp(1.0, 5.0, 0.05).plan_batch(b["x0"], b["zlim"], 0.005, nthreads=8)
ok = ref["status"] == 0
for s in ("0","1"):
    os.environ["CCC_ZMP_STAGE"]=s
    mpc = LinearMpcZmp(1.0, 5.0, 0.05)
    r = mpc.planOnceBatch(b["x0"], b["zlim"], 0.005, want_jerk=True)
    e = np.abs(r["zmp"]-ref["zmp"]); e[~ok]=0
    bad = np.argwhere(e>1e-9)
    print("stage",s,mpc.last_kernel(),"max err",e.max(),"n bad",len(bad),"status nonzero",(r["status"][ok]!=0).sum())
    for i,a in bad[:10]:
        print("   inst",i,"axis",a,"err",e[i,a],"pivots",r["pivots"][i,a],"umax",np.abs(ref["jerk"][i,a]).max(),"oracle iters",ref["iters"][i,a])
