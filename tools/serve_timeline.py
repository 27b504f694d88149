"""Phase timeline of SERVED ticks (closed-loop serving, armed launches): tools/tick2_timeline.py's stamps plus 123 = plant state arrived
in the rollout waves.  Diagnostic build only:  bash tools/build_stamps.sh; DUST_AMD_LIB=tools/_libdust_stamps.so python tools/serve_timeline.py [wait_us]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from dust_amd import Context
from dust_amd import _lib as L

wait_us = float(sys.argv[1]) if len(sys.argv) > 1 else 2000.0
N, S, H, IT = 1024, 128, 30, 5
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
lib = L.load()
lib.dust_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.dust_debug_stamps(c._h, 0, None)  # allocates
state = np.array([3.0, 0.0], np.float32)
c.svmpc_tick(state, 1)
if wait_us >= 0:
    c.serve_start(IT, wait_us)
for _ in range(3000):
    a_seq, _ = c.svmpc_tick(state, IT, want_outputs="action")
    state = bench.pendulum_plant(state, a_seq[0, 0])
# the LAST complete launch's stamps: stop serving WITHOUT another launch overwriting them is not possible (the armed one is cancelled and
# stamps little) - its iteration stamps are absent, the previous launch's remain for those slots; launch-start stamps are the armed one's
st = c.tick_stats()
if wait_us >= 0:
    c.serve_stop()
print(st)
W = N // 4
buf = np.zeros(2048 * 128, np.uint64)
lib.dust_debug_stamps(c._h, -(2048 * 32), buf.ctypes.data_as(C.c_void_p))
tt = buf.reshape(2048, 128).astype(np.int64)
halves = [tt[:W], tt[1024:1024 + W]]
# consecutive launches stamp alternate halves: the half whose launch started EARLIER is the last complete tick (the later one is the
# armed launch serve_stop cancelled - or, unserved, simply the last tick)
starts = [h[:, 120][h[:, 120] > 0].min() for h in halves]
full = halves[int(np.argmin(starts))] if wait_us > 0 else halves[int(np.argmax(starts))]
other_start = max(starts) if wait_us > 0 else None
t = full
t0 = t[:, 120][t[:, 120] > 0].min()
us = lambda x: (x - t0) / 100.0
def show(label, col):
    v = t[:, col]
    v = v[v > 0]
    if len(v):
        print("   %-34s median %7.2f  min %7.2f  max %7.2f" % (label, float(np.median(us(v))), us(v.min()), us(v.max())))
show("launch start", 120)
show("before the first noise", 122)
show("after the initial barrier", 121)
show("plant state arrived (w0)", 123)
names = {0: "iteration start (w0)", 1: "rollouts done (w0)", 2: "theta arrived (w8)", 3: "prior pass done (w8)", 15: "prior pass done (w15)", 4: "after B1",
         5: "weighted sums done (w0)", 7: "score rows published (w8)", 10: "after B4", 12: "after B5: K x score", 14: "after B6"}
for k in (0, 1, IT - 1):
    print("iteration %d:" % k)
    for i in (0, 2, 1, 5, 3, 15, 4, 7, 10, 12, 14):
        show(names[i], 16 * k + i)
print("forward:")
for i, nm in enumerate(["log-density pass done", "log-weights arrived", "end"]):
    show(nm, 16 * IT + i)
print("last stamp of the tick: %.2f us" % us(t.max()))
if other_start is not None:
    print("the NEXT launch (armed, cancelled by serve_stop) started at %.2f us" % us(other_start))
c.close()
