"""Soak test of the in-launch hand-offs: many ticks of the product path at several shapes; every few hundred ticks the state is
compared bitwise with a second context that runs the same ticks through the un-fused kernels (profiling mode).

    python tools/soak.py [ticks]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

# The persistent tick's SHARED prior / Stein tiles (one distance pass in the prior's scaling, persist.hpp tick_pair_shared) differ
# from the un-fused kernels in the last bit of the Stein distances, and 250 closed-loop-free ticks amplify that chaotically; with
# separate tiles the tick is bit-identical to the un-fused path, which is what lets this tool detect a hand-off race.
os.environ.setdefault("DUST_NO_SHARE", "1")

from dust_amd import Context


def run(model, N, S, M, H, ticks, kernel="K1", optimizer="SGD", check_every=250):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(N + H)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    grid = None
    if model == "particle":
        from oracle import grid_4x4_map  # data only

        grid = grid_4x4_map()
    cs = []
    for unfused in (False, True):
        c = Context(model=model, N=N, S=S, M=M, H=H, kernel=kernel, lr=0.5 if optimizer == "SGD" else 0.02, optimizer=optimizer, sigma_a=1.0,
                    sigma_p=1.0, grid=grid, seed=3)
        c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
        c.profile(unfused)
        cs.append(c)
    t0 = time.perf_counter()
    done = 0
    while done < ticks:
        n = min(check_every, ticks - done)
        for c in cs:
            for _ in range(n):
                c.svmpc_tick(state, 5, want_outputs=False)
            c.sync()  # raises if a hand-off spin timed out
        done += n
        a, b = cs[0].get_theta(), cs[1].get_theta()
        assert np.array_equal(a, b), "fused and un-fused paths diverged after %d ticks (max |diff| %g)" % (done, np.abs(a - b).max())
        assert np.isfinite(a).all()
    print("%-9s N=%-5d S=%-4d M=%d H=%-3d %s %s: %d ticks bitwise equal to the un-fused path, %.1f s" % (model, N, S, M, H, kernel, optimizer, ticks, time.perf_counter() - t0), flush=True)
    for c in cs:
        c.close()


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    run("pendulum", 1024, 128, 1, 30, T)
    run("pendulum", 1024, 128, 1, 30, T // 2, kernel="IMQ")
    run("pendulum", 100, 40, 1, 7, T)
    run("pendulum", 100, 40, 1, 7, T // 2, optimizer="Adam")
    run("pendulum", 1024, 128, 1, 30, T // 2, optimizer="Adam")
    run("pendulum", 96, 256, 1, 33, T // 2)
    run("particle", 40, 30, 1, 31, T // 2)
    run("pendulum", 512, 64, 1, 12, T)
