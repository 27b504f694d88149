"""Phase timeline of the persistent tick kernel (svmpc_tick_kernel): every workgroup stamps the 100 MHz wall clock at its phase
boundaries (persist.hpp DUST_TLK, 128 words per workgroup: 16 per SVGD iteration).  Diagnostic build only:

    hipcc <flags of __graft_entry__> -DDUST_STAMPS dust_amd/csrc/dust_amd.hip -o tools/_libdust_stamps.so
    DUST_AMD_LIB=tools/_libdust_stamps.so python tools/tick_timeline.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd import Context
from dust_amd import _lib as L

N, S, H, IT = 1024, 128, 30, 5
rng = np.random.default_rng(0)
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
lib = L.load()
lib.dust_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
lib.dust_debug_stamps(c._h, 0, None)  # allocates
state = np.array([3.0, 0.0], np.float32)
for _ in range(20):
    c.svmpc_tick(state, IT, want_outputs=False)
c.sync()
W = 2048
buf = np.zeros(W * 128, np.uint64)
lib.dust_debug_stamps(c._h, -(W * 32), buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(W, 128).astype(np.int64)
live = t.max(1) > 0
t0 = t[live][t[live] > 0].min()
us = lambda x: (x - t0) / 100.0
n_pair = (N // 32) * 16  # pair workgroups come first in the grid (32-query tiles x 16 key slices), then 2 particles per owner workgroup
idx = np.arange(W)
owners = live & (idx >= n_pair)
pairs = live & (idx < n_pair)
print("workgroups stamped: %d owners, %d pair" % (owners.sum(), pairs.sum()))
names = {0: "start", 1: "actions formed", 2: "rollouts done", 8: "softmax done", 9: "weighted sums done", 3: "prior partials arrived",
         10: "merged", 11: "score row formed", 4: "score published", 5: "next noise drawn", 6: "Stein partials arrived", 12: "phi formed", 7: "theta published"}
order = [0, 1, 2, 8, 9, 3, 10, 11, 4, 5, 6, 12, 7]
for k in range(IT):
    print("iteration %d (median over owners, us since the first stamp of the launch; delta to the previous phase):" % k)
    prev = None
    for i in order:
        v = t[owners, 16 * k + i]
        v = v[v > 0]
        if not len(v):
            continue
        m = float(np.median(us(v)))
        print("   %-24s %7.2f  %s   (min %.2f max %.2f)" % (names[i], m, "" if prev is None else "+%.2f" % (m - prev), us(v.min()), us(v.max())))
        prev = m
    for i in (0, 1, 2):
        v = t[pairs, 16 * k + i]
        v = v[v > 0]
        if len(v):
            print("   pair stamp %d              %7.2f   (min %.2f max %.2f)" % (i, float(np.median(us(v))), us(v.min()), us(v.max())))
kf = IT
v = t[live, 16 * kf:16 * kf + 3]
for i in range(3):
    w = v[:, i][v[:, i] > 0]
    if len(w):
        print("forward stamp %d: median %.2f max %.2f" % (i, float(np.median(us(w))), us(w.max())))
print("last stamp of the launch: %.2f us" % us(t[live].max()))


# placement census (words 126 / 127 of every workgroup: HW_ID and XCC_ID): how the dispatcher spread the two roles over the CUs
t2 = buf.reshape(W, 128)
hw, xcc = t2[:, 126].astype(np.int64), t2[:, 127].astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
import collections
own_per, pair_per = collections.Counter(key[owners]), collections.Counter(key[pairs])
mix = collections.Counter((own_per.get(k_, 0), pair_per.get(k_, 0)) for k_ in set(key[live]))
print("CUs in use: %d; (owner workgroups, pair workgroups) per CU -> number of CUs: %s" % (len(set(key[live])), dict(sorted(mix.items()))))
k = 2
dur = {}
for wg in np.where(owners)[0]:
    a_, b_ = t[wg, 16 * k + 0], t[wg, 16 * k + 2]
    if a_ > 0 and b_ > 0:
        dur.setdefault(own_per[key[wg]], []).append((b_ - a_) / 100.0)
for n_own, v in sorted(dur.items()):
    print("owners on a CU with %d owner workgroups: start -> rollouts done median %.2f us (min %.2f max %.2f, %d workgroups)" % (n_own, np.median(v), min(v), max(v), len(v)))
c.close()
