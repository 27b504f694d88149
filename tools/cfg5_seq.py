"""The cfg5 CONTROL tick alone (Pendulum N=2048, S=128, M=8, H=30, 5 iterations; no filter beside it): python tools/cfg5_seq.py [ticks] [kernel]
- prints ticks/s; under rocprofv3 --kernel-trace the tail of the trace is the steady-state launch sequence (tools/trace_seq.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kernel = sys.argv[2] if len(sys.argv) > 2 else "K1"
N, S, M, H, n_iters = 2048, 128, 8, 30, 5
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2.0 * rng.standard_normal((N, H, 1))).astype(np.float32)
params = (1.0 + 0.1 * rng.standard_normal((n_iters, M, 2))).astype(np.float32)
ctx = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel=kernel, lr=2.0, sigma_a=2.0, sigma_p=2.0, uncertain_params=("length", "mass"), seed=7)
ctx.set_theta(th); ctx.set_prior(mu); ctx.set_a_mat(th)
state = np.array([3.0, 0.0], np.float32)
for _ in range(10):
    ctx.svmpc_tick(state, n_iters, params=params, want_outputs=False)
ctx.sync()
t0 = time.perf_counter()
for _ in range(ticks):
    ctx.svmpc_tick(state, n_iters, params=params, want_outputs=False)
ctx.sync()
el = time.perf_counter() - t0
print("cfg5 control tick / %s: %.0f ticks/s, %.1f us per tick" % (kernel, ticks / el, 1e6 * el / ticks))
ctx.close()
