"""Prior / Stein pass times at large N (cfg4 shape: Particle N=16384, H=40 -> D=80), fused (pairwise_fused.hpp) and unfused.

  python tools/pair_probe.py          all shapes, DUST_PAIR_FUSED=1 and 0
  python tools/pair_probe.py cfg4     the cfg4 shape only (the command profiled for profiles/round2_pair_*)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context
from tools.configs_bench import particle_grid

def run(model, N, H, S=8, kernel="K1", iters=3, alias=True, spread=1.0):
    pend = model == "pendulum"
    da = 1 if pend else 2
    rng = np.random.default_rng(0)
    th = (spread * rng.standard_normal((N, H, da))).astype(np.float32)  # spread 1: almost every kernel value underflows (sparse); 0.02: dense
    c = Context(model=model, N=N, S=S, M=1, H=H, kernel=kernel, sigma_a=1.0, sigma_p=1.5, lr=0.0,  # (lr 0: the particle set keeps its spread)
               
                grid=None if pend else particle_grid(), seed=7)
    c.set_theta(th); c.set_prior(th + 0.1); c.set_a_mat(th)
    state = np.array([3.0, 0.0] if pend else [-9.0, -9.0, 0.0, 0.0], np.float32)
    c.svmpc_optimize(state, 1)
    if alias:
        c.svmpc_forward()
    c.svmpc_optimize(state, 1)
    c.sync()
    c.profile(True)
    c.svmpc_optimize(state, iters)
    c.sync()
    pk = c.profile_get()
    out = {k: 1e3 * v[0] / v[1] for k, v in pk.items()}
    print("%s N=%d D=%d %s alias=%s fused=%s spread=%g: " % (model, N, H * da, kernel, alias, os.environ.get("DUST_PAIR_FUSED", "1"), spread) +
          ", ".join("%s %.0f us" % kv for kv in sorted(out.items())), flush=True)
    c.close()

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
        run("particle", 16384, 40, iters=5, spread=0.02)  # dense: every pair interacts
        sys.exit(0)
    for fused in ("1", "0"):
        os.environ["DUST_PAIR_FUSED"] = fused
        for spread in (0.02, 1.0):
            run("particle", 16384, 40, spread=spread)
            run("pendulum", 16384, 30, spread=spread)
        run("particle", 4096, 40, spread=0.02)
    os.environ.pop("DUST_PAIR_FUSED")
    run("particle", 16384, 20, spread=0.02)
    run("particle", 16384, 40, kernel="IMQ", spread=0.02)
