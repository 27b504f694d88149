"""The two multi-workgroup forms of MPF.optimize (data-polled / counter exchange) on the same inputs: particles after n steps, gradient norms.
    python tools/mpf_compare.py [n_steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd.backend import MpfContext
rng = np.random.default_rng(0)
Mp = 256
x0 = (1.0 + 0.2 * rng.standard_normal((Mp, 2))).astype(np.float32)
st0 = np.array([3.0, 0.0], np.float32); st1 = np.array([3.02, 0.41], np.float32); act = np.array([1.0], np.float32)
res = {}
for mode in ("0", "1"):
    os.environ["DUST_MPF_POLL"] = mode
    m = MpfContext(x0, st0, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-4, init_bw=0.2)
    gn = m.optimize(act, st1, 0.2, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
    res[mode] = (m.get_particles(), gn, m.stats())
    m.close()
a, b = res["0"], res["1"]
print("stats", a[2], b[2])
print("gn counter", a[1][:4], "poll", b[1][:4])
print("max |dx| between forms", np.abs(a[0] - b[0]).max(), " moved (counter) ", np.abs(a[0] - x0).max(), " moved (poll) ", np.abs(b[0] - x0).max())
print(a[0][:3], b[0][:3], x0[:3])
