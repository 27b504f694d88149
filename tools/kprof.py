"""Per-kernel HIP-event timings and (with the -DDUST_STAMPS diagnostic build) in-kernel phase shares.

Development aid; run on the GPU box:   DUST_AMD_LIB=tools/libdust_amd_stamps.so python tools/kprof.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd import _lib as L

NAMES = {0: "rollout", 1: "prior", 2: "stein"}


def stamps(c):
    lib = L.load()
    if not hasattr(lib, "dust_debug_stamps"):
        return None
    out = {}
    for k in NAMES:
        buf = (C.c_ulonglong * 16)()
        lib.dust_debug_stamps(c._h, k, buf)
        v = [int(x) for x in buf]
        out[k] = v
    return out


def run(tag, N=1024, S=128, H=30, M=1, model="pendulum", iters=5, ext=False, reps=20, **kw):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, da))).astype(np.float32)
    grid = None
    if model == "particle":
        from oracle import grid_4x4_map
        grid = grid_4x4_map()
    c = Context(model=model, N=N, S=S, M=M, H=H, kernel=kw.pop("kernel", "K1"), lr=2.0, sigma_a=2.0, sigma_p=2.0, grid=grid, **kw)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    lib = L.load()
    if hasattr(lib, "dust_debug_stamps"):
        lib.dust_debug_stamps(c._h, 0, None)  # first call allocates
    state = np.array([3.0, 0.0] if model == "pendulum" else [-9, -9, 0, 0], np.float32)
    ptr = c.device_noise(iters * S * N * H * da, 5) if ext else None

    def tick():
        if ext:
            c.svmpc_optimize_dev(state, iters, ptr)
        else:
            c.svmpc_tick(state, iters, want_outputs=False)

    for _ in range(3):
        tick()
    c.sync(); c.profile(True)
    for _ in range(reps):
        tick()
    c.sync()
    res = c.profile_get()
    print(tag, " | ".join("%s %.1fus" % (k[:8] + k[-7:], 1e3 * ms / n) for k, (ms, n) in res.items()), flush=True)
    st = stamps(c)
    if st:
        for k, v in st.items():
            d = [v[i + 1] - v[i] for i in range(5) if v[i + 1] and v[i]]
            print("      %-8s block0 phase cycles:" % NAMES[k], d, "total", (v[5] - v[0]) if v[5] else None)
    c.close()


if __name__ == "__main__":
    run("base philox   ")
    run("external eps  ", ext=True)
    run("H=1           ", H=1)
    run("N=4096        ", N=4096, reps=5)
    run("particle 256  ", model="particle", N=256, S=64, M=4, H=40, uncertain_params=None, reps=5)
