import numpy as np, sys, os
sys.path.insert(0,'/root/repo')
from bnpc_amd import _lib, hostkernels
from bnpc_amd.model import TMIN, TMAX
kt=hostkernels.table(); rng=np.random.RandomState(0); sd=np.array([.1,.25,.5])
for G in (3,10):
    M=1000
    old=np.clip(rng.uniform(size=(G,M)),TMIN,TMAX).astype(np.float32)
    n1=rng.randint(0,500,(G,M)).astype(np.int32); n0=rng.randint(0,500,(G,M)).astype(np.int32)
    for i in range(6):
        _lib.mh_batch(kt,old,n1,n0,sd,TMIN,TMAX,.01,.2,.25,.25,False,False,want_prior=True,threads=16)
