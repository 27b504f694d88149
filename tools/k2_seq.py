"""The cfg2 / K2 tick alone (no profiling mode behind it): python tools/k2_seq.py [ticks] - prints ticks/s; under rocprofv3 --kernel-trace the
tail of the trace is the steady-state launch sequence of the K2 tick (tools/trace_seq.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kernel = sys.argv[2] if len(sys.argv) > 2 else "K2"
N, S, H, n_iters = 1024, 128, 30, 5
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2.0 * rng.standard_normal((N, H, 1))).astype(np.float32)
ctx = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel=kernel, lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=7)
ctx.set_theta(th); ctx.set_prior(mu); ctx.set_a_mat(th)
state = np.array([3.0, 0.0], np.float32)
for _ in range(30):
    ctx.svmpc_tick(state, n_iters, want_outputs=False)
ctx.sync()
t0 = time.perf_counter()
for _ in range(ticks):
    ctx.svmpc_tick(state, n_iters, want_outputs=False)
ctx.sync()
el = time.perf_counter() - t0
print("cfg2 / %s: %.0f ticks/s, %.1f us per tick" % (kernel, ticks / el, 1e6 * el / ticks))
ctx.close()
