import numpy as np, sys
sys.path.insert(0,'.')
from autorally_amd import capi, synthetic as S
from oracle import oracle as O
from tests.helpers import noise_for, warm_U, rel_err
K,T=2048,100
cfg=S.make_config(K,T,track='ring')
eps=noise_for(cfg); U0=warm_U(cfg)
orc=O.Oracle(cfg,nthreads=8)
ref=orc.compute_control(cfg['start_state'],U0,np.zeros(4,np.float32),eps)
c_ref,V,crash=orc.rollouts(cfg['start_state'],U0,eps[0])
orc0=O.Oracle(cfg,fma_mode=0,nthreads=8)
c_nf,_,crash_nf=orc0.rollouts(cfg['start_state'],U0,eps[0])
sol=capi.Solver(cfg); sol.set_control_seq(U0); sol.set_noise(eps); sol.compute_control(cfg['start_state'])
got=sol.get_results()
err=rel_err(got['costs'],ref['costs'])
bad=np.where(err>1e-4)[0]
print('gpu-vs-oracle bad',bad, 'maxerr excluding bad', err[err<=1e-4].max())
for k in bad: print(k,'gpu',got['costs'][k],'ref',ref['costs'][k],'crash_ref',crash[k],'w_ref',ref['w'][k],'w_gpu',got['w'][k])
e2=rel_err(c_nf,c_ref); b2=np.where(e2>1e-4)[0]
print('oracle fma vs nofma bad',b2,[ (c_nf[k],c_ref[k]) for k in b2])
print('dU',np.abs(got['U']-ref['U']).max(),'traj',got['traj_cost'],ref['traj_cost'])
# downstream consistency: oracle weights+reduction+SG fed with the GPU's costs
w,b,eta,tc=orc.weights(got['costs'])
U=orc.savgol(orc.weighted_reduction(w,eta,got_V:=sol.get_applied_controls()),np.zeros(4,np.float32))
print('downstream dU',np.abs(U-got['U']).max(),'tc',tc,got['traj_cost'])
