"""Development aid (GPU box): the persistent one-launch tick (persist.hpp) against the launch-per-iteration path.

  python tools/persist_check.py            # bitwise comparison on several shapes + timing at cfg2
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context


TOL = 5e-4  # whole-tick chains amplify rounding (softmax of costs O(1e3)); stage-wise parity is pinned by tests/


def make(model, N, S, H, M=1, seed=0, **kw):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, da))).astype(np.float32)
    grid = None
    if model == "particle":
        from oracle import grid_4x4_map

        grid = grid_4x4_map()
    sig = 2.0 if model == "pendulum" else 5.0
    c = Context(model=model, N=N, S=S, M=M, H=H, kernel=kw.pop("kernel", "K1"), lr=kw.pop("lr", 2.0 if model == "pendulum" else 100.0),
                sigma_a=sig, sigma_p=sig, grid=grid, seed=77, **kw)
    c.set_theta(th)
    c.set_prior(mu)
    c.set_a_mat(th)
    return c, rng


def state_of(c):
    return dict(theta=c.get_theta(), a_mat=c.get_a_mat(), costs=c.get_costs(), score=c.get_score(), phi=c.get_phi(),
                ll=c.get_log_weights()[0], lp=c.get_log_weights()[1], prior=c.get_prior()[0], mix=c.get_prior()[1])


def run(persist, model, N, S, H, iters, ticks, ext, **kw):
    if persist:
        os.environ.pop("DUST_NO_PERSIST", None)
    else:
        os.environ["DUST_NO_PERSIST"] = "1"
    c, rng = make(model, N, S, H, **kw)
    st = np.array([3.0, 0.0], np.float32) if model == "pendulum" else np.array([-9, -9, 0, 0], np.float32)
    outs = []
    da = 1 if model == "pendulum" else 2
    erng = np.random.default_rng(5)
    for t in range(ticks):
        eps = erng.standard_normal((iters, S, N, H, da)).astype(np.float32) if ext else None
        a, pw = c.svmpc_tick(st, iters, eps)
        outs.append((a.copy(), pw.copy()))
    c.sync()
    s = state_of(c)
    c.close()
    return outs, s


def compare(tag, *args, **kw):
    o1, s1 = run(True, *args, **kw)
    o0, s0 = run(False, *args, **kw)
    bad = []
    for t, ((a1, p1), (a0, p0)) in enumerate(zip(o1, o0)):
        if np.abs(a1 - a0).max() > TOL * max(np.abs(a0).max(), 1e-30):
            bad.append("a_seq[t%d] maxdiff %.3g" % (t, np.abs(a1 - a0).max()))
        d = np.abs(p1 - p0).max() / max(np.abs(p0).max(), 1e-30)
        if d > 1e-5:
            bad.append("pw[t%d] rel %.3g" % (t, d))
    worst = 0.0
    for k in s1:
        if not np.array_equal(s1[k], s0[k]):
            d = np.abs(s1[k].astype(np.float64) - s0[k]).max() / max(np.abs(s0[k]).max(), 1e-30)
            worst = max(worst, d)
            if d < TOL:
                continue  # shared-distance pair tiles / forward's 256-lane reductions: rounding-level differences
            bad.append("%s rel %.3g" % (k, d))
    print("%-44s %s" % (tag, ("OK (bitwise)" if worst == 0 else "OK (max rel %.2g)" % worst) if not bad else "DIFF: " + "; ".join(bad)), flush=True)
    return not bad


def timing(persist, steps=300, want=False):
    if persist:
        os.environ.pop("DUST_NO_PERSIST", None)
    else:
        os.environ["DUST_NO_PERSIST"] = "1"
    c, _ = make("pendulum", 1024, 128, 30)
    st = np.array([3.0, 0.0], np.float32)
    for _ in range(30):
        c.svmpc_tick(st, 5, want_outputs=want)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.svmpc_tick(st, 5, want_outputs=want)
    c.sync()
    el = time.perf_counter() - t0
    c.close()
    return steps / el


if __name__ == "__main__":
    ok = True
    if "--time-only" not in sys.argv:
        ok &= compare("pendulum N=16 S=8 H=10 ext", "pendulum", 16, 8, 10, 2, 3, True)
        ok &= compare("pendulum N=64 S=32 H=15 ext", "pendulum", 64, 32, 15, 3, 3, True)
        ok &= compare("pendulum N=64 S=32 H=15 philox", "pendulum", 64, 32, 15, 3, 3, False)
        ok &= compare("pendulum N=96 S=128 H=30 philox adam", "pendulum", 96, 128, 30, 2, 3, False, optimizer="Adam", lr=0.1)
        ok &= compare("pendulum N=1024 S=128 H=30 ext", "pendulum", 1024, 128, 30, 5, 2, True)
        ok &= compare("pendulum N=1024 S=128 H=30 philox", "pendulum", 1024, 128, 30, 5, 3, False)
        ok &= compare("pendulum N=256 S=256 H=30 philox IMQ", "pendulum", 256, 256, 30, 2, 2, False, kernel="IMQ")
        ok &= compare("particle N=128 S=64 H=20 ext weighted", "particle", 128, 64, 20, 2, 3, True, weighted_prior=True)
        ok &= compare("particle N=256 S=64 H=30 philox mean-roll", "particle", 256, 64, 30, 1, 3, False, roll_strategy="mean")
    for want in (False, True):
        tp = timing(True, want=want)
        tl = timing(False, want=want)
        print("cfg2 ticks/s want_outputs=%s: persistent %.0f   launch-per-iteration %.0f" % (want, tp, tl), flush=True)
    sys.exit(0 if ok else 1)
