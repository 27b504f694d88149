"""The dual loop at cfg5 size on one GPU (Pendulum N=2048, S=128, M=8, H=30, 5 SVGD iterations; filter 256 particles x 20 steps,
`mpf_bandwidth: null`), CLOSED: every period's action steps a host plant and the new state feeds the filter update and the next tick.
  (a) through the C ABI piece by piece, the way dust_amd/controllers/dual.py composes them unfused: filter particles to the host + host
      Silverman rule, dust_mpf_optimize, dust_mpf_prior_sample to the host, dust_svmpc_tick with host samples;
  (b) dust_dual_tick: one call per period (Silverman's rule and the dynamics samples on the device);
  (c), (d) the same two through the Python mirror classes (DualSVMPC(fused=False / True)).
    python tools/dual_loop_time.py [periods]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from dust_amd.backend import MpfContext
from dust_amd.inference.mpf import silvermans_rule

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, S, M, H, K, Mp = 2048, 128, 8, 30, 5, 256
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2.0 * rng.standard_normal((N, H, 1))).astype(np.float32)
x0 = (1.0 + 0.2 * rng.standard_normal((Mp, 2))).astype(np.float32)


def plant(st, a):
    thd = np.float32(np.clip(st[1] + 0.05 * (14.7 * np.sin(st[0]) + 3.0 * np.clip(a, -2, 2)), -8, 8))
    return np.array([st[0] + thd * 0.05, thd], np.float32)


def make():
    c = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, uncertain_params=("length", "mass"), seed=7)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    m = MpfContext(x0, np.array([3.0, 0.0], np.float32), model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3)
    return c, m


for name in ("pieces", "dust_dual_tick"):
    c, m = make()
    st, prev = np.array([3.0, 0.0], np.float32), None
    for t in range(T + 20):
        if t == 20:
            c.sync(); t0 = time.perf_counter()
        if name == "pieces":
            if prev is not None:
                bw = silvermans_rule(m.get_particles().reshape(-1, 1).astype(np.float64))
                m.optimize(prev, st, float(bw), 20)
            params = m.prior_sample(K * M, t + 1).reshape(K, M, 2)
            a_seq, pw = c.svmpc_tick(st, K, None, params)
        else:
            a_seq, pw, bw = c.dual_tick(m, st, prev, K, 20, None, t + 1)
        prev = a_seq[0].copy()
        st = plant(st, float(a_seq[0, 0]))
    el = time.perf_counter() - t0
    print("cfg5 closed dual loop, C ABI, %-15s %7.1f us per period (%5.0f periods/s)" % (name + ":", 1e6 * el / T, T / el), flush=True)
    c.close(); m.close()
