import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
sys.argv=[sys.argv[0]]
from tools.persist_check import make
c,_=make("pendulum",1024,128,30)
st=np.array([3.0,0.0],np.float32)
for _ in range(30): c.svmpc_tick(st,5,want_outputs=False)
c.sync()
for steps in (100,):
    t0=time.perf_counter(); hs=[]
    for _ in range(steps):
        h0=time.perf_counter(); c.svmpc_tick(st,5,want_outputs=False); hs.append(time.perf_counter()-h0)
    th=time.perf_counter()-t0
    c.sync(); el=time.perf_counter()-t0
    hs=np.array(hs)*1e6
    print("steps %4d: %.1f us/tick total, host enqueue %.1f us/tick (median call %.1f, max %.1f)"%(steps, el/steps*1e6, th/steps*1e6, np.median(hs), hs.max()), flush=True)
