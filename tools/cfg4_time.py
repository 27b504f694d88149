"""cfg4 (Particle N = 16384, S = 64, M = 4, H = 40, one SVGD iteration per tick) on one GPU: ms per tick for K1 and IMQ (python tools/cfg4_time.py; under rocprofv3 --kernel-trace --stats it gives the per-launch time of the large-set pairwise kernels)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bench import particle_grid
from dust_amd import Context
N, S, M, H = 16384, 64, 4, 40
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 2)).astype(np.float32)
th = (mu + rng.standard_normal((N, H, 2))).astype(np.float32)
for kern in ("K1", "IMQ"):
    c = Context(model="particle", N=N, S=S, M=M, H=H, kernel=kern, lr=100.0 if kern == "K1" else 1.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0, uncertain_params=("mass",), grid=particle_grid(), seed=3)
    c.set_theta(th); c.set_prior(th); c.set_a_mat(th)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    pr = (1.0 + 0.1 * rng.standard_normal((1, M, 1))).astype(np.float32)
    for _ in range(5): c.svmpc_tick(st, 1, params=pr, want_outputs=False)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(20): c.svmpc_tick(st, 1, params=pr, want_outputs=False)
    c.sync()
    print(kern, "%.3f ms per tick" % ((time.perf_counter() - t0) / 20 * 1e3), "theta checksum %.6f" % float(np.abs(c.get_theta()).sum()), flush=True)
    c.close()
