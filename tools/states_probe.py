"""Rollout kernel at cfg3 size with and without stored states (whole-line form + second pass, binary16 fallback), Pendulum stored form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from tools.configs_bench import particle_grid

def run(model, N, S, M, H, store, f16=False, reps=3):
    pend = model == "pendulum"
    da = 1 if pend else 2; ds = 2 if pend else 4
    rng = np.random.default_rng(0)
    th = rng.standard_normal((N, H, da)).astype(np.float32)
    up = None; params = None
    if M > 1:
        up = ("length", "mass") if pend else ("mass",)
        params = (1.0 + 0.1 * rng.standard_normal((M, len(up)))).astype(np.float32)
    c = Context(model=model, N=N, S=S, M=M, H=H, kernel="K1", sigma_a=2.0 if pend else 1.0, sigma_p=1.0, uncertain_params=up,
                grid=None if pend else particle_grid(), seed=7)
    c.set_theta(th); c.set_prior(th); c.set_a_mat(th)
    state = np.array([3.0, 0.0] if pend else [-9.0, -9.0, 0.0, 0.0], np.float32)
    eps = rng.standard_normal((S, N, H, da)).astype(np.float32)
    c.profile(True)
    for _ in range(reps + 1):
        c.likelihood_sample(state, eps, params, store_states=store, store_f16=f16)
    c.sync()
    pk = c.profile_get()
    ms, n = pk["rollout_kernel"]
    us = 1e3 * ms / n
    if "states_kernel" in pk:
        ms2, n2 = pk["states_kernel"]
        print("   whole-line states kernel %.1f us + second pass %.1f us" % (1e3 * ms2 / n2, us))
        us += 1e3 * ms2 / n2
    R = M * S * N
    b_states = (2 if f16 else 4) * R * (H + 1) * ds if store else 0
    b_alg = 4 * (S * N * H * da + 2 * N * H * da + S * N) + b_states
    print("%s N=%d S=%d M=%d H=%d store=%s f16=%s: %.1f us, %.2f GB algorithmic -> %.0f GB/s (%.1f %% of 8 TB/s)" %
          (model, N, S, M, H, store, f16, us, b_alg / 1e9, b_alg / us / 1e3, 100 * b_alg / us / 1e3 / 8000), flush=True)
    c.close()

if __name__ == "__main__":
    run("particle", 4096, 64, 64, 40, False)
    run("particle", 4096, 64, 64, 40, True)
    run("particle", 4096, 64, 64, 40, True, f16=True)
    run("pendulum", 8192, 128, 8, 30, True)
    run("pendulum", 8192, 128, 8, 30, True, f16=True)
    run("pendulum", 8192, 128, 8, 30, False)
