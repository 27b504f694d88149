"""Is the tick host-bound?  Compare the time to ENQUEUE K ticks with the time until they have finished (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
N, S, H = 1024, 128, 30
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32); th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
st = np.array([3.0, 0.0], np.float32)
for _ in range(20): c.svmpc_tick(st, 5, want_outputs=False)
c.sync()
K = 200
t0 = time.perf_counter()
for _ in range(K): c.svmpc_tick(st, 5, want_outputs=False)
t1 = time.perf_counter()
c.sync()
t2 = time.perf_counter()
print("enqueue %.1f us/tick, total %.1f us/tick" % (1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K))
