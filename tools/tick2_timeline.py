"""Phase timeline of the owner-computes tick kernel (tick2.hpp T2_TL stamps, 100 MHz wall clock, 128 words per workgroup: 8 per
SVGD iteration, forward behind them, 120/121 launch start / after the initial barrier).  Diagnostic build only:

    F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -Iinclude -Idust_amd/csrc -DDUST_STAMPS"
    hipcc $F -c dust_amd/csrc/dust_amd.hip -o /tmp/a.o & hipcc $F -c dust_amd/csrc/tick2.hip -o /tmp/b.o; wait
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/a.o /tmp/b.o -ldl -o tools/_libdust_stamps.so
    DUST_AMD_LIB=tools/_libdust_stamps.so python tools/tick2_timeline.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd import _lib as L

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
for _ in range(300):
    c.svmpc_tick(state, IT, want_outputs=False)
c.sync()
print(c.tick_stats())
W = N // 4
buf = np.zeros(2048 * 128, np.uint64)
lib.dust_debug_stamps(c._h, -(2048 * 32), buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(2048, 128).astype(np.int64)[:W]
t0 = t[:, 120][t[:, 120] > 0].min()
us = lambda x: (x - t0) / 100.0
names = {0: "iteration start (w0)", 1: "rollouts done (w0)", 2: "theta arrived (w8)", 3: "prior pass done (w8)", 15: "prior pass done (w15)", 4: "after B1",
         5: "weighted sums done (w0)", 7: "score rows published (w8)", 8: "score arrivals seen (w10)", 9: "next noise drawn (w0)",
         11: "Stein pass done (w8)", 6: "Stein pass done (w15)", 10: "after B4", 12: "after B5: K x score", 13: "theta rows published (w8)", 14: "after B6"}
order = [0, 2, 1, 5, 3, 15, 4, 7, 9, 11, 6, 8, 10, 12, 13, 14]
def show(label, col):
    v = t[:, col]
    v = v[v > 0]
    if len(v):
        print("   %-34s median %7.2f  min %7.2f  max %7.2f" % (label, float(np.median(us(v))), us(v.min()), us(v.max())))
show("launch start", 120)
show("before the first noise", 122)
show("after the initial barrier", 121)
for k in range(IT):
    print("iteration %d:" % k)
    for i in order:
        show(names[i], 16 * k + i)
print("forward:")
for i, nm in enumerate(["log-density pass done", "log-weights arrived", "end"]):
    show(nm, 16 * IT + i)
print("last stamp: %.2f us" % us(t.max()))
c.close()
