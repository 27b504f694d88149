"""MpfContext.optimize(20 steps) and (2 steps) per particle count: time per call and per step, which kernel served the calls.
    MPS=128,256 python tools/mpf_time.py   (DUST_MPF_GRID=0|1, DUST_MPF_POLL=0 select the kernel)  -> profiles/round3_mpf_time.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd.backend import MpfContext
rng = np.random.default_rng(0)
for Mp in [int(v) for v in os.environ.get("MPS", "128,256,512,1024").split(",")]:
    x0 = (1.0 + 0.2 * rng.standard_normal((Mp, 2))).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    m = MpfContext(x0, state, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3)
    act = np.zeros(1, np.float32)
    for _ in range(20): m.optimize(act, state, 0.1, 20)
    t0 = time.perf_counter()
    for _ in range(200): m.optimize(act, state, 0.1, 20)
    t20 = 1e6 * (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200): m.optimize(act, state, 0.1, 2)
    t2 = 1e6 * (time.perf_counter() - t0) / 200
    print("Mp=%d: %.1f us per optimize(20 steps), %.1f (2 steps) -> %.2f us per step, %s" % (Mp, t20, t2, (t20 - t2) / 18, m.stats()), flush=True)
