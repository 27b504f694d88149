"""cfg5 on one GPU as a loop: Pendulum N=2048, S=128, M=8, H=30, 5 SVGD iterations per tick (asynchronous) + MPF 256 x 20 (synchronous)
per tick; prints us per joint tick and which kernels served the calls.  MPF=0: the tick alone.
    python tools/cfg5_loop.py [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from dust_amd.backend import MpfContext
N, S, M, H, n_iters = 2048, 128, 8, 30, 5
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2.0 * rng.standard_normal((N, H, 1))).astype(np.float32)
up = ("length", "mass")
params = (1.0 + 0.1 * rng.standard_normal((n_iters, M, 2))).astype(np.float32)
ctx = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel=os.environ.get("KERNEL", "K1"), lr=2.0, sigma_a=2.0, sigma_p=2.0, uncertain_params=up, seed=7)
ctx.set_theta(th); ctx.set_prior(mu); ctx.set_a_mat(th)
state = np.array([3.0, 0.0], np.float32)
x0 = (1.0 + 0.2 * rng.standard_normal((256, 2))).astype(np.float32)
mpf = MpfContext(x0, state, model="pendulum", uncertain_params=up, obs_std=0.1, lr=1e-3) if os.environ.get("MPF", "1") == "1" else None
act = np.zeros(1, np.float32)
def tick():
    ctx.svmpc_tick(state, n_iters, params=params, want_outputs=False)
    if mpf is not None: mpf.optimize(act, state, 0.1, 20)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(20): tick()
ctx.sync()
t0 = time.perf_counter()
for _ in range(T): tick()
ctx.sync()
print("cfg5: %.1f us/tick, %s, mpf %s" % (1e6 * (time.perf_counter() - t0) / T, ctx.tick_stats(), mpf.stats() if mpf is not None else None))
