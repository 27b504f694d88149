"""pairwise_far.hpp on the cfg4 workload of bench.py: per tick, the time and the share of (query tile, key chunk) units the
pre-pass proves to be exact zeros - with the pre-pass (default) and without it (DUST_FAR=0; DUST_PROBE_ONLY=1 / 0 runs one of the two).
python tools/far_probe.py [ticks]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd import Context, _lib as L

lib = L.load()
lib.dust_debug_far_units.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
lib.dust_debug_far_logp.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
n_ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
c4 = dict(bench.CFG3, n_iters=1) if os.environ.get("DUST_PROBE_CFG") == "3" else bench.CFG4  # (DUST_PROBE_CFG=3: the cfg3 shape)
for far in ((os.environ["DUST_PROBE_ONLY"],) if os.environ.get("DUST_PROBE_ONLY") else ("1", "0")):
    os.environ["DUST_FAR"] = far
    mu4, theta4 = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
    one = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                  uncertain_params=("mass",), grid=bench.particle_grid(), device=0, seed=1234)
    one.set_theta(theta4); one.set_prior(mu4); one.set_a_mat(theta4)
    p4 = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
    st4 = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    rows = []
    for k in range(n_ticks):
        one.sync()
        t0 = time.perf_counter()
        one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False)
        one.sync()
        dt = time.perf_counter() - t0
        out = (C.c_longlong * 2)()
        lib.dust_debug_far_units(one._h, out)
        out2 = (C.c_longlong * 2)()
        lib.dust_debug_far_logp(one._h, out2)
        rows.append((k, dt * 1e6, out[0], out[1], out2[0], out2[1]))
    print("DUST_FAR=%s" % far)
    for k, us, f, u, f2, u2 in rows:
        print("  tick %2d  %8.1f us   far units %7d of %7d (%.3f)   log-p blocks %7d of %7d (%.3f)" % (k, us, f, u, f / u if u else 0.0, f2, u2, f2 / u2 if u2 else 0.0))
    th = one.get_theta()
    print("  theta checksum %.9g" % float(np.float64(th).sum()), flush=True)
    one.close()
