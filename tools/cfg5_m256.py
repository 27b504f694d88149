import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from oracle import Oracle
N, S, M, H = 64, 128, 256, 30
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32); th = (mu + rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel="IMQ", uncertain_params=("length", "mass"), lr=2.0, sigma_a=2.0, sigma_p=2.0)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
st = np.array([3.0, 0.0], np.float32)
eps = rng.standard_normal((S, N, H, 1)).astype(np.float32)
params = rng.uniform(0.6, 1.3, (M, 2)).astype(np.float32)
costs = c.likelihood_sample(st, eps, params)
o = Oracle(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=("length", "mass"))
ref = o.rollout_cost(st, o.sample_actions(th, eps, np.full(1, 2.0, np.float32)), params)
print("M=256 rollouts vs oracle: max rel err %.2e" % (np.abs(costs - ref) / np.abs(ref)).max())
c.close()
# the cfg5 shape with M = 256: ticks/s
N = 2048
mu = rng.standard_normal((N, H, 1)).astype(np.float32); th = (mu + rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel="IMQ", uncertain_params=("length", "mass"), lr=2.0, sigma_a=2.0, sigma_p=2.0)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
p5 = rng.uniform(0.6, 1.3, (5, M, 2)).astype(np.float32)
for _ in range(3): c.svmpc_tick(st, 5, params=p5, want_outputs=False)
c.sync(); t0 = time.perf_counter()
for _ in range(10): c.svmpc_tick(st, 5, params=p5, want_outputs=False)
c.sync(); el = (time.perf_counter() - t0) / 10
print("cfg5 with M = 256 (N=2048, S=128, H=30, 5 iterations, IMQ): %.2f ms per tick = %.1f ticks/s; tick paths %s" % (el * 1e3, 1 / el, c.tick_stats()))
