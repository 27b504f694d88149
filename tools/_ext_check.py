import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from dust_amd import Context
from dust_amd import _lib as L
N,S,H=1024,128,30
rng=np.random.default_rng(0)
mu=rng.standard_normal((N,H,1)).astype(np.float32); th=(mu+2*rng.standard_normal((N,H,1))).astype(np.float32)
c=Context(model="pendulum",N=N,S=S,M=1,H=H,kernel="K1",lr=2.0,sigma_a=2.0,sigma_p=2.0)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
lib=L.load()
if hasattr(lib,"dust_debug_stamps") and os.environ.get("STAMPS_ON"): lib.dust_debug_stamps(c._h,0,None)
state=np.array([3.0,0.0],np.float32)
ptr=c.device_noise(5*S*N*H,5)
if os.environ.get("PROF"): c.profile(True)
for _ in range(int(os.environ.get("REPS","3"))):
    c.svmpc_optimize_dev(state,5,ptr)
c.sync()
np.save(sys.argv[1], c.get_theta())
print("ok", float(np.abs(c.get_theta()).mean()))
