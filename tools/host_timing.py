"""Host-side cost of a control tick at cfg2: open loop, synchronised, with outputs read back, closed loop with a host plant
(WITH_TORCH=1 imports torch first: the first ~200 output ticks are then 2.5x slower, a one-off)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if os.environ.get("WITH_TORCH"): import torch
sys.argv=[sys.argv[0]]
from tools.persist_check import make
c,_=make("pendulum",1024,128,30)
st=np.array([3.0,0.0],np.float32)
for _ in range(30): c.svmpc_tick(st,5,want_outputs=False)
c.sync()
def loop(name, fn, n=200):
    for _ in range(10): fn()
    c.sync()
    t0=time.perf_counter()
    for _ in range(n): fn()
    c.sync()
    print("%-40s %.1f us/tick" % (name, (time.perf_counter()-t0)/n*1e6), flush=True)
loop("open loop (no outputs, no sync)", lambda: c.svmpc_tick(st,5,want_outputs=False))
loop("no outputs + sync per tick", lambda: (c.svmpc_tick(st,5,want_outputs=False), c.sync()))
for rep in range(4): loop("outputs (D2H + sync) per tick #%d" % rep, lambda: c.svmpc_tick(st,5,want_outputs=True))
loop("sync only", lambda: c.sync())
os.environ["DUST_NO_PERSIST"]="1"
loop("launch-per-iteration: outputs per tick", lambda: c.svmpc_tick(st,5,want_outputs=True))
loop("launch-per-iteration: open loop", lambda: c.svmpc_tick(st,5,want_outputs=False))
os.environ.pop("DUST_NO_PERSIST")
import bench
stt = st.copy()
def cl():
    global stt
    a_seq, _ = c.svmpc_tick(stt, 5, want_outputs=True)
    stt = bench.pendulum_plant(stt, a_seq[0, 0])
loop("closed loop with bench's plant", cl)
import math
def plant2(s, u, dt=0.05, g=9.8, m=1.0, l=1.0):
    th, thd = float(s[0]), float(s[1])
    u = min(max(float(u), -2.0), 2.0)
    thd = thd + dt * (-3.0 * g / (2.0 * l) * math.sin(th + math.pi) + 3.0 * u / (m * l * l))
    thd = min(max(thd, -8.0), 8.0)
    s[0] = th + thd * dt; s[1] = thd
    return s
def cl2():
    a_seq, _ = c.svmpc_tick(stt, 5, want_outputs=True)
    plant2(stt, a_seq[0, 0])
loop("closed loop, python-float plant", cl2)
