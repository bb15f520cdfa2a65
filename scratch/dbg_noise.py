import numpy as np, sys
sys.path.insert(0,'.')
from autorally_amd import capi, synthetic as S
from oracle import oracle as O
K,T=2048,100
cfg=S.make_config(K,T)
sol=capi.Solver(cfg); sol.seed(1234,0)
a=sol.generate_noise()
ref=O.generate_noise(1234,0,K,T)
bad=np.argwhere(a.view(np.uint32)!=ref.view(np.uint32))
print(len(bad))
for k,t,j in bad[:20]:
    print(k,t,j,repr(a[k,t,j]),repr(ref[k,t,j]), a[k,t], ref[k,t])
print('t hist', np.bincount(bad[:,1],minlength=T))
print('max abs diff', np.max(np.abs(a-ref)))
