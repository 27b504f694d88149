"""Soak of the owner-computes tick (tick2.hpp): many ticks at several shapes with the hand-off time-out / abort counters checked and,
every `check_every` ticks, ONE tick compared with a clone that runs the same tick through the tiled one-launch kernel
(DUST_NO_TICK2 is read per call): a hand-off that lets a consumer run early shows up as a tick that disagrees with its replay.
(The two kernels sum in different orders, so whole runs cannot be compared bitwise as tools/soak.py does for the tiled form:
the comparison is per tick, from identical state, at the tolerance of tests/test_gpu_tick2.py.)

    python tools/soak_tick2.py [ticks]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context


def elemerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + np.sqrt(np.mean(b * b)) + 1e-300)))


def run(model, N, S, M, H, ticks, kernel="K1", optimizer="SGD", check_every=500):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(N + H)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    kw = {}
    if model == "particle":
        from oracle import grid_4x4_map  # data only

        kw["grid"] = grid_4x4_map()
    sig = 2.0 if da == 1 else 5.0
    c = Context(model=model, N=N, S=S, M=M, H=H, kernel=kernel, lr=2.0 if da == 1 else 5.0, optimizer=optimizer, sigma_a=sig, sigma_p=sig,
                seed=3, **kw)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    t0 = time.perf_counter()
    done, worst = 0, 0.0
    while done < ticks:
        n = min(check_every, ticks - done)
        for _ in range(n - 1):
            c.svmpc_tick(state, 5, want_outputs=False)
        c.sync()  # raises if a hand-off spin timed out
        os.environ["DUST_NO_TICK2"] = "1"  # (development switches are read once per context, when it is created: the clone is a new context)
        try:
            twin = c.clone()
        finally:
            os.environ.pop("DUST_NO_TICK2", None)
        a_seq, pw = c.svmpc_tick(state, 5)
        b_seq, qw = twin.svmpc_tick(state, 5)
        assert twin.tick_stats()["tick2"] == 0, twin.tick_stats()
        e = max(elemerr(c.get_theta(), twin.get_theta()), elemerr(c.get_phi(), twin.get_phi()))
        worst = max(worst, e)
        # (the weights are exp(log_w - logsumexp(log_w)) in fp32, as in the reference (svmpc.py:140): with Particle costs of 1e4-1e5 one ulp
        #  of the normaliser is 0.8 % - sums of 0.97-1.03 are the formula's, in 1 % of the ticks of the Particle shapes, on every path)
        assert np.isfinite(c.get_theta()).all() and abs(float(pw.sum()) - 1.0) < 5e-2
        # (one tick from identical state: summation order x the softmax over costs of O(1e3) - 1e-3 is usual late in a run; a consumer that
        #  ran ahead of its producer shows up as O(1))
        assert e < 5e-2, "tick %d: owner-computes and tiled tick disagree from identical state: %g" % (done + n, e)
        twin.close()
        done += n
    st = c.tick_stats()
    assert st["replayed"] == 0, st
    print("%s N=%d S=%d M=%d H=%d %s %s: %d ticks, %.1f us/tick, paths %s, worst per-tick disagreement with the tiled kernel %.2e"
          % (model, N, S, M, H, kernel, optimizer, ticks, 1e6 * (time.perf_counter() - t0) / ticks, st, worst), flush=True)
    c.close()


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    run("pendulum", 1024, 128, 1, 30, T)
    run("pendulum", 1024, 128, 1, 30, T, kernel="IMQ", optimizer="Adam")
    run("pendulum", 512, 64, 1, 20, T)
    run("pendulum", 96, 64, 4, 20, T // 2)
    run("particle", 128, 64, 4, 16, T // 2)
    run("particle", 256, 64, 1, 16, T // 2, kernel="IMQ")
