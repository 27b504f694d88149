"""One rank's compute time per tick of the strong-scaled cfg4 run (Particle N = 16384 over G ranks), with the REAL data flow: all G
shards live in this process on one GPU (LocalComm: the all-gathers are slice copies), the set evolves tick by tick exactly as in
the G-GPU run, and rank 0's four phases are timed one by one (stream synchronisation before and after each; the other ranks run
untimed in between).  Collectives excluded - comm_probe / the bench's `comm_us_per_tick` measure those.
    python tools/shard_emul.py [G,G,..] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd.parallel import DeviceShard, LocalComm

Gs = [int(g) for g in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
n_ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
c4 = bench.CFG4
for G in Gs:
    mu, theta = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
    common = dict(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                  uncertain_params=("mass",), grid=bench.particle_grid(), seed=1234)
    params = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    shards = [DeviceShard(common, r, G) for r in range(G)]  # (every context on torch's stream: the slice copies of LocalComm are ordered with the kernels)
    for sh in shards:
        sh.set_state(theta, mu)
    comm = LocalComm()
    E = shards[0].shard_elems
    rows = []
    for k in range(n_ticks):
        ph = {}
        def timed(name, fn):
            shards[0].sync()
            t0 = time.perf_counter()
            fn(shards[0])
            shards[0].sync()
            ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t0) * 1e6
            for sh in shards[1:]:
                fn(sh)
        for it in range(c4["n_iters"]):
            timed("local_score", lambda sh: sh.local_score(st, None, params[it]))
            comm.all_gather_inplace(shards, "score_all", E)
            timed("apply_phi", lambda sh: sh.apply_phi())
            comm.all_gather_inplace(shards, "theta_all", E)
        timed("forward_local", lambda sh: sh.forward_local())
        comm.all_gather_inplace(shards, "lw_all", shards[0].n_loc)
        timed("forward_finish", lambda sh: sh.forward_finish(False))
        rows.append(ph)
    tail = rows[n_ticks // 2:]
    mean = {k: float(np.mean([r[k] for r in tail])) for k in tail[0]}
    print("cfg4 G=%d n_local=%d: rank 0 compute per tick %.0f us (ticks %d..%d: %s)" %
          (G, c4["N"] // G, sum(mean.values()), n_ticks // 2, n_ticks - 1, {k: round(v) for k, v in mean.items()}), flush=True)
    for sh in shards:
        sh.ctx.close()
