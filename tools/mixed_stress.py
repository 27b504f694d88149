"""One process, three host threads, one GPU: a cfg2 controller ticking through the owner-computes kernel (needs every CU) beside two
MPF contexts updating through the multi-workgroup filter kernel (64 spinning workgroups each).  Expected: no error, no hang; ticks that
find the chip occupied abort at their start barrier and are replayed, filter calls that cannot start fall back to one workgroup.
    python tools/mixed_stress.py"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from dust_amd.backend import MpfContext
rng = np.random.default_rng(0)
N, S, H = 1024, 128, 30
mu = rng.standard_normal((N, H, 1)).astype(np.float32); th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
st = np.array([3.0, 0.0], np.float32)
x0 = (1.0 + 0.2 * rng.standard_normal((256, 2))).astype(np.float32)
mpfs = [MpfContext(x0, st, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-4, init_bw=0.2) for _ in range(2)]
err = []
def ticks():
    try:
        for t in range(3000):
            a, pw = c.svmpc_tick(st, 5)
            assert np.isfinite(a).all() and abs(float(pw.sum()) - 1) < 1e-3
    except Exception as e: err.append(("tick", repr(e)))
def filt(m):
    try:
        for t in range(1500):
            gn = m.optimize(np.array([0.5], np.float32), st, 0.2, 20)
            assert np.isfinite(gn).all()
    except Exception as e: err.append(("mpf", repr(e)))
t0 = time.perf_counter()
ths = [threading.Thread(target=ticks)] + [threading.Thread(target=filt, args=(m,)) for m in mpfs]
[t.start() for t in ths]; [t.join() for t in ths]
print("mixed stress: %.1f s, errors %s, tick paths %s, mpf %s" % (time.perf_counter() - t0, err, c.tick_stats(), [m.stats() for m in mpfs]))
