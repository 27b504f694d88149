"""World-1 cost of the C-side sharded tick (cfg2): unsharded persistent tick / unsharded launch-per-iteration tick / sharded tick
through a one-rank RCCL communicator (DUST_COMM_FORCE=1: every ncclAllGather is issued)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dust_amd import Context

def run(env, sharded, steps=300):
    for k in ("DUST_NO_PERSIST", "DUST_COMM_FORCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.default_rng(0)
    N, H = 1024, 30
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
    kw = dict(model="pendulum", N=N, S=128, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=1)
    c = Context(shard_offset=0, shard_size=N, **kw) if sharded else Context(**kw)
    if sharded:
        c.comm_init(Context.comm_unique_id(), 0, 1)
    c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
    st = np.array([3.0, 0.0], np.float32)
    for _ in range(30):
        c.svmpc_tick(st, 5, want_outputs=False)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c.svmpc_tick(st, 5, want_outputs=False)
    c.sync()
    us = (time.perf_counter() - t0) / steps * 1e6
    c.close()
    return us

if __name__ == "__main__":
    a = run({}, False)
    b = run({"DUST_NO_PERSIST": "1"}, False)
    d = run({"DUST_COMM_FORCE": "1"}, True)
    print("cfg2 us/tick: persistent %.1f | launch-per-iteration (graph) %.1f | sharded world-1 with RCCL %.1f  -> 12 all-gathers + sharded kernels cost %.1f us over the graph path"
          % (a, b, d, d - b))
