import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import ctypes
from dust_amd import _lib as L
_l=ctypes.CDLL(L.LIB_PATH)
for k in list(L.SYMBOLS):
    if not hasattr(_l,k): del L.SYMBOLS[k]
from dust_amd import Context
N,S,H=1024,128,30
rng=np.random.default_rng(0)
mu=rng.standard_normal((N,H,1)).astype(np.float32); th=(mu+2*rng.standard_normal((N,H,1))).astype(np.float32)
c=Context(model="pendulum",N=N,S=S,M=1,H=H,kernel="K1",lr=2.0,sigma_a=2.0,sigma_p=2.0)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
state=np.array([3.0,0.0],np.float32)
ptr=c.device_noise(8*S*N*H,5)
for _ in range(3):
    print(os.environ.get("DUST_AMD_LIB"), "avg us", round(c.profile_rollout(state, ptr, 8, 400)*1e3,2), flush=True)
