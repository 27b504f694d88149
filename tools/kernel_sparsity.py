#!/usr/bin/env python3
"""How sparse is the Stein kernel matrix in fp32 at the bench workloads?  Runs T ticks (device noise), fetches the particles and
counts the off-diagonal pairs whose K1 value exp(-d^2 / (2 ln(2)^2)) is not exactly 0 in fp32 (d^2 < ~84), and the pairs whose
prior weight exp(-d^2 / (2 sigma_p^2)) is above 2^-126 relative to the self term."""
import sys
import numpy as np

sys.path.insert(0, ".")
from dust_amd import Context  # noqa: E402


def report(tag, c, state, iters, ticks_list):
    done = 0
    for T in ticks_list:
        for _ in range(T - done):
            c.svmpc_tick(state, iters)
        done = T
        th = c.get_theta().reshape(c.N, -1).astype(np.float64)
        n = min(c.N, 2048)
        x = th[:n]
        d2 = ((x * x).sum(1)[:, None] + (th * th).sum(1)[None, :] - 2 * x @ th.T)
        d2[np.arange(n), np.arange(n)] = np.inf
        ell2 = np.log(2.0) ** 2
        nz = (d2 / (2 * ell2) < 87.3).sum()
        rows = ((d2 / (2 * ell2) < 87.3).sum(1) > 0).sum()
        print("%s after %4d ticks: min d2 %.1f  median d2 %.1f | K1 nonzero off-diagonal pairs %d of %d (rows with any: %d of %d) | spread per coord %.2f"
              % (tag, T, d2.min(), np.median(d2[np.isfinite(d2)]), nz, n * c.N - n, rows, n, th.std(0).mean()), flush=True)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    N, S, H = 1024, 128, 30
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    c.set_theta((mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)); c.set_prior(mu); c.set_a_mat(mu)
    report("cfg2", c, np.array([3.0, 0.0], np.float32), 5, [1, 10, 100, 1000, 5000])
    c.close()
    from oracle import grid_4x4_map  # test infrastructure (the map only)
    N, S, H, M = 16384, 64, 40, 4
    c = Context(model="particle", N=N, S=S, M=M, H=H, kernel="K1", lr=100.0, sigma_a=5.0, sigma_p=5.0, seed=1, grid=grid_4x4_map(),
                uncertain_params=("mass",))
    mu = rng.standard_normal((N, H, 2)).astype(np.float32)
    c.set_theta((mu + 5 * rng.standard_normal((N, H, 2))).astype(np.float32)); c.set_prior(mu); c.set_a_mat(mu)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    done = 0
    for T in [1, 20, 100]:
        for _ in range(T - done):
            params = (2.0 + 0.1 * rng.standard_normal((1, M, 1))).astype(np.float32)
            c.svmpc_tick(st, 1, params=params)
        done = T
        report("cfg4", c, st, 1, [])
        th = c.get_theta().reshape(N, -1).astype(np.float64)
        x = th[:1024]
        d2 = ((x * x).sum(1)[:, None] + (th * th).sum(1)[None, :] - 2 * x @ th.T)
        d2[np.arange(1024), np.arange(1024)] = np.inf
        print("cfg4 after %3d ticks: min d2 %.1f median %.1f, K1 nonzero off-diagonal pairs %d, prior-weight nonzero fraction %.3f, spread %.2f"
              % (T, d2.min(), np.median(d2[np.isfinite(d2)]), (d2 / (2 * np.log(2.0) ** 2) < 87.3).sum(), (d2 / 50.0 < 87.3).mean(), th.std(0).mean()), flush=True)
    c.close()
