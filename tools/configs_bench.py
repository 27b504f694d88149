"""Ticks/s and per-kernel times of the SURVEY 8(d) configurations on ONE MI355X (cfg4 / cfg5 are 8-GPU configurations in
the survey; here the whole problem runs on one device, which bounds the per-GPU work from above).

    python tools/configs_bench.py [out.json]

Development / reporting aid; bench.py stays the contract benchmark (cfg2).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd.backend import MpfContext


def particle_grid():
    # 220 x 220 cells of 0.1 m with the demo's 4 x 4 block obstacle pattern scaled up (synthetic occupancy)
    g = np.zeros((220, 220), np.float32)
    for bx in range(4):
        for by in range(4):
            x0, y0 = 30 + bx * 45, 30 + by * 45
            g[x0:x0 + 18, y0:y0 + 18] = 1.0
    return g


def run(name, ticks, warm, model, N, S, M, H, n_iters, kernel="K1", mpf=None, lr=None, **kw):
    pend = model == "pendulum"
    da = 1 if pend else 2
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + (2.0 if pend else 1.0) * rng.standard_normal((N, H, da))).astype(np.float32)
    up = None
    params = None
    if M > 1:
        up = ("length", "mass") if pend else ("mass",)
        P = len(up)
        params = (1.0 + 0.1 * rng.standard_normal((n_iters, M, P))).astype(np.float32)
    ctx = Context(model=model, N=N, S=S, M=M, H=H, kernel=kernel, lr=(2.0 if pend else 100.0) if lr is None else lr,
                  sigma_a=2.0 if pend else 1.0, sigma_p=2.0 if pend else 1.0, uncertain_params=up,
                  grid=None if pend else particle_grid(), seed=7, **kw)
    ctx.set_theta(th)
    ctx.set_prior(mu)
    ctx.set_a_mat(th)
    state = np.array([3.0, 0.0] if pend else [-9.0, -9.0, 0.0, 0.0], np.float32)
    if kw.get("control_type") == "velocity":
        state = state[:2]  # (a two-state model: particle.py:41-48)
    m = None
    if mpf:
        x0 = (1.0 + 0.2 * rng.standard_normal((mpf["Mp"], 2))).astype(np.float32)
        m = MpfContext(x0, state, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3)
    act = np.zeros(da, np.float32)

    def tick():
        ctx.svmpc_tick(state, n_iters, params=params, want_outputs=False)
        if m is not None:
            m.optimize(act, state, 0.1, mpf["steps"])

    for _ in range(warm):
        tick()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(ticks):
        tick()
    ctx.sync()
    el = time.perf_counter() - t0
    ctx.profile(True)
    for _ in range(max(2, ticks // 10)):
        ctx.svmpc_tick(state, n_iters, params=params, want_outputs=False)
    ctx.sync()
    pk = {k: round(1e3 * ms / n, 1) for k, (ms, n) in ctx.profile_get().items()}
    ctx.profile(False)
    res = dict(config=name, model=model, N=N, S=S, M=M, H=H, n_iters=n_iters, kernel=kernel, mpf=mpf, ticks=ticks,
               ticks_per_s=ticks / el, ms_per_tick=1e3 * el / ticks, rollouts_per_s=ticks / el * n_iters * N * S * M,
               per_kernel_us_unfused=pk, rollout_algorithmic_GBps=None)
    if "rollout_kernel" in pk:
        res["rollout_algorithmic_GBps"] = ctx.rollout_bytes() / (pk["rollout_kernel"] * 1e-6) / 1e9
    ctx.close()
    if m is not None:
        m.close()
    print(json.dumps(res), flush=True)
    return res


def main():
    out = []
    only = os.environ.get("DUST_CONFIGS_ONLY")  # development: substring of the rows to run
    if only:
        global run
        real_run = run
        # a single row is timed right after process start: 10 ticks on cold clocks read 20-30 % low (round 6: an A/B of the noisy cfg3 row
        # came out backwards that way) - half a second of cfg2 ticks first
        wc = None if os.environ.get("DUST_CONFIGS_NO_WARM") else Context(model="pendulum", N=1024, S=128, M=1, H=30, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
        if wc is not None:
            w0 = np.zeros((1024, 30, 1), np.float32)
            wc.set_theta(w0); wc.set_prior(w0); wc.set_a_mat(w0)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.5:
                for _ in range(50):
                    wc.svmpc_tick(np.array([3.0, 0.0], np.float32), 5, want_outputs=False)
                wc.sync()
            wc.close()

        exact = os.environ.get("DUST_CONFIGS_EXACT")  # the row's whole name ("cfg3" alone also names the noisy / velocity rows)

        def run(name, *a, **k):  # noqa: F811
            return real_run(name, *a, **k) if (name == only if exact else only in name) else dict(config=name, skipped=True)

    out.append(run("cfg1", 300, 30, "pendulum", 32, 128, 1, 15, 1))
    out.append(run("cfg2 (bench.py)", 300, 30, "pendulum", 1024, 128, 1, 30, 5))
    out.append(run("cfg2 / K2 (iid_mp)", 100, 10, "pendulum", 1024, 128, 1, 30, 5, kernel="K2"))
    out.append(run("cfg2 / IMQ", 300, 30, "pendulum", 1024, 128, 1, 30, 5, kernel="IMQ"))
    out.append(run("cfg3", 20, 3, "particle", 4096, 64, 64, 40, 1))
    # Particle(deterministic=False, noise_std=0.1) - the reference's constructor default with real noise - and velocity control:
    # particle_general.hpp
    out.append(run("cfg3 noisy (control-channel noise 0.1)", 10, 2, "particle", 4096, 64, 64, 40, 1, deterministic=False, noise_std=(0.1, 0.1)))
    out.append(run("cfg3 velocity control", 10, 2, "particle", 4096, 64, 64, 40, 1, control_type="velocity"))
    out.append(run("cfg4 on one GPU", 20, 3, "particle", 16384, 64, 4, 40, 1))
    out.append(run("cfg5 on one GPU (K1, M=8, MPF 256 x 20)", 50, 5, "pendulum", 2048, 128, 8, 30, 5, mpf=dict(Mp=256, steps=20)))
    out.append(run("cfg5 on one GPU (IMQ, M=8, MPF 256 x 20)", 50, 5, "pendulum", 2048, 128, 8, 30, 5, kernel="IMQ", mpf=dict(Mp=256, steps=20)))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
