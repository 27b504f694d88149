"""Launch timeline of the one-launch SVGD iteration (svgd_iter_kernel): every workgroup stamps the 100 MHz wall clock at its
role's boundaries (4 words each).  Diagnostic build only:

    hipcc ... -DDUST_STAMPS -o tools/libdust_amd_stamps.so;  DUST_AMD_LIB=tools/libdust_amd_stamps.so python tools/iter_timeline.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd import _lib as L


def main(N=1024, S=128, H=30, kernel="K1"):
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel=kernel, lr=2.0, sigma_a=2.0, sigma_p=2.0)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    lib = L.load()
    lib.dust_debug_stamps(c._h, 0, None)  # allocates
    state = np.array([3.0, 0.0], np.float32)
    for _ in range(3):
        c.svmpc_optimize(state, 2)
    c.sync()
    tiles, JS = (N + 31) // 32, 16
    P, R = tiles * JS, N // 2
    U = (N * H + 255) // 256
    G = 2 * P + R + U
    roles = [("prior", 0, P, ["entry", "-", "-", "arrived"]), ("rollout", P, P + R, ["entry", "at prior wait", "tail done", "published"]),
             ("stein", P + R, 2 * P + R, ["entry", "theta-only part done", "saw scores", "arrived"]),
             ("update", 2 * P + R, G, ["entry", "admitted", "-", "done"])]
    for rep in range(3):
        c.svmpc_optimize(state, 1)
        c.sync()
        buf = (C.c_ulonglong * (4 * G))()
        lib.dust_debug_stamps(c._h, -G, buf)
        v = np.array(buf, dtype=np.uint64).reshape(G, 4).astype(np.int64)
        t0 = v[:, 0].min()
        print("launch %d (us after the first workgroup's entry; min / median / max over the role's workgroups)" % rep)
        for name, a, b, labels in roles:
            for k, lab in enumerate(labels):
                if lab == "-":
                    continue
                x = (v[a:b, k] - t0) * 0.01
                print("  %-8s %-22s %6.2f %6.2f %6.2f" % (name, lab, x.min(), np.median(x), x.max()))
    c.close()


if __name__ == "__main__":
    main()
