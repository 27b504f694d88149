import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from dust_amd import Context
from oracle import grid_4x4_map
N,S,M,H=256,64,1,17; da=2; rng=np.random.default_rng(3)
mu = rng.standard_normal((N, H, da)).astype(np.float32); th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
state=np.array([-9.0,-9.0,0,0],np.float32)
X=th.reshape(N,-1).astype(np.float64); Y=mu.reshape(N,-1).astype(np.float64)
d2=((X[:,None,:]-Y[None,:,:])**2).sum(-1); lg=-0.5*d2; lg-=lg.max(1,keepdims=True); r=np.exp(lg); r/=r.sum(1,keepdims=True)
gp=(r[:,:,None]*(Y[None,:,:]-X[:,None,:])).sum(1)
c = Context(model="particle", N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, grid=grid_4x4_map(), seed=11)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
c.likelihood_sample(state)
phi,gl,gpr=c.svmpc_phi()
e=np.abs(gpr.reshape(N,-1)-gp); print("stage-wise (unfused prior + prior_finish): max err per col>=30", e[:,30:].max(0), "cols<30", e[:,:30].max())
gl=gl.reshape(N,-1)
c.close()
for unfused in (False,True):
    c = Context(model="particle", N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, grid=grid_4x4_map(), seed=11)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th); c.profile(unfused)
    c.svmpc_optimize(state,1); c.sync()
    sc=c.get_score().reshape(N,-1)
    e=np.abs((sc-gl)-gp); print("unfused" if unfused else "fused", "err per col>=30:", e[:,30:].max(0), "cols<30 max", e[:,:30].max(), "bad rows", np.nonzero(e[:,32]>1e-3)[0][:20])
    c.close()
