"""cfg2 / K2, 1500 ticks in the default form (bandwidth role inside the prior + rollout launch, phi on the row-major particles) and with
DUST_K2_FORM=0 (the launches of rounds 1-5): the particles and bandwidths must stay BIT-IDENTICAL tick after tick (checked every 100).
    python tools/k2_soak.py [ticks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
N, S, H = 1024, 128, 30
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2.0 * rng.standard_normal((N, H, 1))).astype(np.float32)
ctxs = []
for form in ("2", "0"):
    os.environ["DUST_K2_FORM"] = form
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K2", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=7)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    ctxs.append(c)
state = np.array([3.0, 0.0], np.float32)
for t in range(ticks):
    outs = [c.svmpc_tick(state, 5 if t % 7 else 4) for c in ctxs]
    state = np.array([np.cos(0.01 * t) * 3.0, np.sin(0.02 * t)], np.float32)
    if t % 100 == 99:
        a, b = (c.get_theta() for c in ctxs)
        ha, hb = (c.get_bandwidths() for c in ctxs)
        ok = np.array_equal(a, b) and np.array_equal(ha, hb) and np.array_equal(outs[0][0], outs[1][0]) and np.isfinite(a).all()
        print("tick %4d: identical %s  |theta| %.4f  h[0] %.6f" % (t + 1, ok, float(np.abs(a).mean()), float(ha[0])), flush=True)
        if not ok:
            sys.exit(1)
print("ok")
