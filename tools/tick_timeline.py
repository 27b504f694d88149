"""Timeline of the persistent one-launch tick (persist.hpp svmpc_tick_kernel): every workgroup stamps the 100 MHz wall clock at
its phase boundaries (16 slots per iteration).  Diagnostic build only:

    hipcc ... -DDUST_STAMPS -o tools/libdust_amd_stamps.so;  DUST_AMD_LIB=tools/libdust_amd_stamps.so python tools/tick_timeline.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd import _lib as L

PAIR = {0: "theta(k) seen", 3: "  prior: staged", 4: "  prior: pass A done", 5: "  prior: softmax done", 6: "  prior: pass B done",
        7: "  prior: stores issued", 1: "prior tile arrived", 8: "  stein: staged", 9: "  stein: pass A done", 10: "  stein: repulsion done",
        11: "  stein: scores seen", 2: "stein tile arrived"}
OWN = {0: "iter start", 1: "actions ready", 2: "rollouts done", 8: "  softmax done", 9: "  weighted sums done", 3: "prior partials seen",
       10: "  merge barrier", 11: "  score stored", 4: "score published", 5: "next noise drawn", 6: "stein partials seen",
       12: "  partials loaded", 7: "theta published"}
FWD_PAIR = {0: "theta(n) seen", 1: "logp tile arrived"}
FWD_OWN = {0: "logp partials seen", 1: "all log-weights seen", 2: "done"}


def main(N=1024, S=128, H=30, kernel="K1", iters=5, show=(2,)):
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel=kernel, lr=2.0, sigma_a=2.0, sigma_p=2.0)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    lib = L.load()
    lib.dust_debug_stamps(c._h, 0, None)  # allocates
    state = np.array([3.0, 0.0], np.float32)
    for _ in range(5):
        c.svmpc_tick(state, iters, want_outputs=False)
    c.sync()
    tiles, JS = (N + 31) // 32, 16
    P, R = tiles * JS, N // 2
    G = P + R
    c.svmpc_tick(state, iters, want_outputs=False)
    c.sync()
    buf = (C.c_ulonglong * (128 * G))()
    lib.dust_debug_stamps(c._h, -(32 * G), buf)
    v = np.array(buf, dtype=np.uint64).reshape(G, 128).astype(np.int64)
    t0 = v[P:, 0].min()
    print("us after the first owner's start; min / median / max over the role's workgroups")
    marks = []
    for k in range(iters + 1):
        for name, a, b, labels in (("pair", 0, P, PAIR if k < iters else FWD_PAIR), ("owner", P, G, OWN if k < iters else FWD_OWN)):
            for j, lab in labels.items():
                x = (v[a:b, 16 * k + j] - t0) * 0.01
                marks.append((k, name, lab, x.min(), float(np.median(x)), x.max()))
    for k, name, lab, lo, med, hi in marks:
        if k in show or k == iters or not lab.startswith("  "):
            print("  k=%d %-6s %-26s %7.2f %7.2f %7.2f" % (k, name, lab, lo, med, hi))
    # placement census: which workgroups share a CU (HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13; XCC_ID bits 3:0)
    hw, xcc = v[:, 126], v[:, 127] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    from collections import Counter
    mix = Counter()
    for key in np.unique(cu):
        blocks = np.nonzero(cu == key)[0]
        mix[(int((blocks < P).sum()), int((blocks >= P).sum()))] += 1
    print("CUs by (pair workgroups, owner workgroups) resident:", dict(mix), " distinct CUs:", len(np.unique(cu)))
    own_done = (v[P:, 16 * 2 + 2] - t0) * 0.01
    for key in list(np.unique(cu))[:0]:
        pass
    # rollouts-done time of owners grouped by how many owner workgroups share their CU
    per_cu_owner = {key: int(((cu == key) & (np.arange(G) >= P)).sum()) for key in np.unique(cu)}
    for n_own in sorted(set(per_cu_owner.values())):
        sel = np.array([per_cu_owner[k] == n_own for k in cu[P:]])
        if sel.any():
            print("  owners on CUs with %d owner workgroups: rollouts done (k=2) median %.2f max %.2f  (n=%d)" % (n_own, np.median(own_done[sel]), own_done[sel].max(), sel.sum()))
    c.close()


if __name__ == "__main__":
    main()
