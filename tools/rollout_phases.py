"""In-kernel phase shares of rollout_kernel at the cfg4 / cfg3 shapes (diagnostic build with s_memtime stamps):
    bash tools/build_stamps.sh && DUST_AMD_LIB=tools/_libdust_stamps.so python tools/rollout_phases.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd import Context, _lib as L

lib = L.load()
PH = ["stage action tile (noise)", "rollouts", "softmax weights", "eta / flags", "weighted sums -> score, a_mat"]
for tag, cfg in (("cfg4", bench.CFG4), ("cfg3", bench.CFG3)):
    mu, th = bench.synth(cfg["N"], cfg["H"], 2, spread=1.0)
    c = Context(model="particle", N=cfg["N"], S=cfg["S"], M=cfg["M"], H=cfg["H"], kernel="K1", lr=1.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                uncertain_params=("mass",), grid=bench.particle_grid(), seed=1234)
    c.set_theta(th); c.set_prior(th); c.set_a_mat(th)
    lib.dust_debug_stamps(c._h, 0, None)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    p = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((1, cfg["M"], 1))).astype(np.float32)
    for _ in range(4):
        c.svmpc_tick(st, 1, params=p, want_outputs=False)
    c.sync()
    buf = (C.c_ulonglong * 16)()
    lib.dust_debug_stamps(c._h, 0, buf)
    v = [int(x) for x in buf]
    d = [v[i + 1] - v[i] for i in range(5)]
    tot = v[5] - v[0]
    print(tag, "block 0 of rollout_kernel, s_memtime ticks (100 MHz):", " | ".join("%s %d (%.0f%%)" % (PH[i], d[i], 100.0 * d[i] / tot) for i in range(5)), "total", tot, flush=True)
    c.close()
