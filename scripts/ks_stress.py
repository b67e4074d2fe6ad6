"""Development stress of the state-space kernel of LinearMpcZmp (csrc/zmp_stage.inc) against the CPU oracle: 4 horizons x 12
seeds x 3000 instances through CCC_ZMP_STAGE=1 (KS + the exact kernel on its hand-over list).  Round 6, one MI355X: 288000
QPs, worst |d ZMP| 1.6e-14 m at 2 s horizons / 1.4e-12 m at 3 s, worst relative jerk error 7.5e-10, no QP unsolved.
usage (GPU box): python scripts/ks_stress.py"""
import numpy as np, os, sys, time
sys.path.insert(0,'.')
os.environ["CCC_ZMP_STAGE"]="1"
from centroidalcontrolcollection_amd import LinearMpcZmp, fixtures as fx
from oracle import oracle
worst=0; worstj=0; t0=time.time(); nq=0; bad=0
for N,dt in ((64,2.0/64),(100,0.02),(150,2.0/150),(100,0.03)):
    T=N*dt
    mpc=LinearMpcZmp(1.0,T,dt); o=oracle.LinearMpcZmp(1.0,T,dt)
    for seed in range(1000,1012):
        b=fx.make_zmp_batch(3000,N,dt,seed=seed)
        ref=o.plan_batch(b["x0"],b["zlim"],0.005,nthreads=16)
        r=mpc.planOnceBatch(b["x0"],b["zlim"],0.005,want_jerk=True)
        ok=(ref["status"]==0)
        e=np.abs(r["zmp"]-ref["zmp"])[ok].max()
        sc=np.maximum(1.0,np.abs(ref["jerk"]).max(axis=-1,keepdims=True))
        ej=(np.abs(r["jerk"]-ref["jerk"])/sc)[ok].max()
        worst=max(worst,e); worstj=max(worstj,ej); nq+=2*ok.sum(); bad+=int((r["status"][ok]!=0).sum())
        if e>1e-9 or ej>1e-7: print("MISMATCH",N,dt,seed,e,ej)
    print("N",N,"dt",dt,"worst zmp err",worst,"worst jerk rel err",worstj,"QPs",nq,"non-ok",bad,"t",time.time()-t0,flush=True)
