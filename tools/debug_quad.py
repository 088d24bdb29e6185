import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge, bench
from oracle import cloudy_oracle as O
import test_gpu_numerical as T, test_gpu_parity as P
pkg=ge.load_package()
for dist_types,kname,nq in (([1,1,1],"constant",4),([1,1],"hydro",2)):
    par,op,okf=T.numerical_case(pkg,O,dist_types,kname,nq)
    mom=P.mixed_moments(dist_types,1500,seed=100+7*len(dist_types)+nq)
    got=T.run_numerical(pkg,par,mom)
    want,scale=O.rhs_coal_numerical_batch(op,okf,nq,mom,with_scale=True)
    prm=O.update_dist_batch(op,mom)
    err=np.abs(got-want)/np.maximum(scale,1e-300)
    idx=np.argsort(-np.nanmax(err,axis=0))[:6]
    np.set_printoptions(precision=6,linewidth=200)
    for i in idx:
        print(kname,nq,"parcel",i,"err",err[:,i]); print("  params",prm[:,i]); print("  got ",got[:,i]); print("  want",want[:,i]); print("  scale",scale[:,i])
