import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from peps_amd import capi, fermion
from oracle import fermion as of
from oracle.bmps import BMPSTruncateParams
from oracle.graded import GT
L,D,chi=8,6,24
st=fermion.random_even_state(L,L,D,seed=11)
gts=[[[GT(st.tensors[r][c][s][...,None], list(st.par[r][c])+[np.array([int(st.nf[s])])],[-1,1,1,-1,-1]) for s in range(2)] for c in range(L)] for r in range(L)]
fs=of.FermionSITPS(gts)
rng=np.random.default_rng(7)
cfgs=np.stack([rng.permutation(np.r_[np.zeros(32,dtype=int),np.ones(32,dtype=int)]).reshape(L,L) for _ in range(4)])
tp=BMPSTruncateParams.SVD(chi,chi,0.0)
ref=np.array([fs.amplitude(c,tp) for c in cfgs])
for dt,name in ((capi.F64,"f64"),(capi.F32,"f32")):
    import os
    for env in ({}, {"PEPSGPU_NO_RANK_ADAPT":"1"}):
        os.environ.pop("PEPSGPU_NO_RANK_ADAPT",None); os.environ.update(env)
        ctx=capi.Context(L,L,D,8,chi,dtype=dt,max_walkers=4); ctx.state_upload(st.extended_flat(D))
        amp=fermion.evaluate_amplitude(ctx,st,cfgs)
        print(name,env,np.abs(amp/ref-1))
# exact reference with larger chi to see truncation sensitivity
ref2=np.array([fs.amplitude(c,BMPSTruncateParams.SVD(48,48,0.0)) for c in cfgs[:2]])
print("chi24 vs chi48 oracle", np.abs(ref[:2]/ref2-1))
