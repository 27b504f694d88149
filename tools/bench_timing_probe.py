import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as entry
entry.build()
import torch
import bench
from dust_amd import Context
w = bench.WORKLOAD
mu, theta = bench.synth(w["N"], w["H"], 1)
ctx = Context(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"], sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=0, seed=1234)
ctx.set_theta(theta); ctx.set_prior(mu); ctx.set_a_mat(theta)
st = np.array([3.0, 0.0], np.float32)
tick = lambda: ctx.svmpc_tick(st, 5, want_outputs=False)
n, t0 = 0, time.perf_counter()
while n < 5 or time.perf_counter() - t0 < 0.1:
    for _ in range(25):
        tick(); n += 1
    ctx.sync()
print("warm", n)
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tick()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("rep %d: enqueue %.1f us, +sync %.1f us, +torch sync %.1f us -> %.1f us/tick" % (rep, (t1-t0)*1e6, (t2-t1)*1e6, (t3-t2)*1e6, (t3-t0)/20*1e6))
# pure overhead of the synchronisation calls on an idle stream
for rep in range(3):
    t0 = time.perf_counter(); ctx.sync(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("idle: ctx.sync %.1f us, torch sync %.1f us" % ((t1-t0)*1e6, (t2-t1)*1e6))
# back-to-back 20-tick regions without any host work in between
res = []
for rep in range(8):
    t0 = time.perf_counter()
    for _ in range(20):
        tick()
    ctx.sync()
    res.append((time.perf_counter() - t0) / 20 * 1e6)
print("back-to-back 20-tick regions:", " ".join("%.1f" % r for r in res))
t0 = time.perf_counter()
for _ in range(300):
    tick()
ctx.sync()
print("300 ticks: %.1f us/tick" % ((time.perf_counter() - t0) / 300 * 1e6))
# bench.py's exact sequence, repeated: warm-up in batches for 0.1 s, then 20 timed ticks
def warm(n_min=5, secs=0.1):
    n, t0 = 0, time.perf_counter()
    while n < n_min or time.perf_counter() - t0 < secs:
        for _ in range(25):
            tick(); n += 1
        ctx.sync()
    return n
for rep in range(4):
    n = warm()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tick()
    ctx.sync()
    torch.cuda.synchronize()
    print("warm %d ticks, then 20 timed: %.1f us/tick" % (n, (time.perf_counter() - t0) / 20 * 1e6))
for rep in range(3):
    n = warm(secs=0.3)
    t0 = time.perf_counter()
    for _ in range(20):
        tick()
    ctx.sync()
    print("warm %d ticks (0.3 s), then 20 timed, no torch sync: %.1f us/tick" % (n, (time.perf_counter() - t0) / 20 * 1e6))
