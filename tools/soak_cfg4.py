import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd import Context
c4 = bench.CFG4
for kern, n in (("K1", 1200), ("IMQ", 100)):
    mu4, theta4 = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
    one = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel=kern, lr=100.0 if kern == "K1" else 1.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                  uncertain_params=("mass",), grid=bench.particle_grid(), device=0, seed=1234)
    one.set_theta(theta4); one.set_prior(mu4); one.set_a_mat(theta4)
    p4 = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
    st4 = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    t0 = time.perf_counter()
    for k in range(n):
        want = (k % 100 == 99)
        r = one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=want)
        if want:
            a_seq, pw = r
            th = one.get_theta()
            print(kern, "tick", k, "finite", bool(np.isfinite(th).all() and np.isfinite(a_seq).all()), "sum pw %.6f" % float(pw.sum()), "%.2f ms/tick" % ((time.perf_counter() - t0) / (k + 1) * 1e3), flush=True)
    one.close()
print("soak ok")
