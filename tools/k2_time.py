"""cfg2 with the K2 (iid_mp, per-dimension median bandwidth) kernel: us per tick (python tools/k2_time.py [K2 K2shared])."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from dust_amd import Context
w = bench.WORKLOAD
mu, theta = bench.synth(w["N"], w["H"], 1)
for kern in (sys.argv[1:] or ["K2"]):
    try:
        ctx = Context(model="pendulum", N=w["N"], S=w["S"], M=1, H=w["H"], kernel=kern, lr=w["lr"], alpha=w["alpha"], sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=0, seed=1234)
    except Exception as e:
        print(kern, "skip", e); continue
    ctx.set_theta(theta); ctx.set_prior(mu); ctx.set_a_mat(theta)
    st = np.array([3.0, 0.0], np.float32)
    for _ in range(300): ctx.svmpc_tick(st, 5, want_outputs=False)
    ctx.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(300): ctx.svmpc_tick(st, 5, want_outputs=False)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
    print(kern, "%.1f us per tick = %.0f ticks/s" % (best, 1e6 / best), flush=True)
    ctx.close()
