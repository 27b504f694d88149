"""One rank's tick of the strong-scaled cfg4 run, on the AGED set, as the GPU sees it: the set is aged with all G shards in this process
(tools/shard_emul.py's data flow), then, `reps` times, rank 0 runs ONE whole tick alone (no host synchronisation between its
phases), timed between two events, is put back, and a joint tick follows.  Under `rocprofv3 --kernel-trace` the last ticks of the trace are rank 0's and nothing
else: tools/trace_seq.py prints their kernel sequence with gaps.
    python tools/rank_trace.py G [age_ticks] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd.parallel import DeviceShard, LocalComm

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
age = int(sys.argv[2]) if len(sys.argv) > 2 else 150
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
c4 = bench.CFG4
mu, theta = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
common = dict(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
              uncertain_params=("mass",), grid=bench.particle_grid(), seed=1234)
params = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
shards = [DeviceShard(common, r, G) for r in range(G)]
for sh in shards:
    sh.set_state(theta, mu)
comm = LocalComm()
E = shards[0].shard_elems
for k in range(age):
    for it in range(c4["n_iters"]):
        for sh in shards:
            sh.local_score(st, None, params[it])
        comm.all_gather_inplace(shards, "score_all", E)
        for sh in shards:
            sh.apply_phi()
        comm.all_gather_inplace(shards, "theta_all", E)
    for sh in shards:
        sh.forward_local()
    comm.all_gather_inplace(shards, "lw_all", shards[0].n_loc)
    for sh in shards:
        sh.forward_finish(False)
for sh in shards:
    sh.sync()
r0 = shards[0]
import torch


def joint_tick():
    for it in range(c4["n_iters"]):
        for sh in shards:
            sh.local_score(st, None, params[it])
        comm.all_gather_inplace(shards, "score_all", E)
        for sh in shards:
            sh.apply_phi()
        comm.all_gather_inplace(shards, "theta_all", E)
    for sh in shards:
        sh.forward_local()
    comm.all_gather_inplace(shards, "lw_all", shards[0].n_loc)
    for sh in shards:
        sh.forward_finish(False)


# Rank 0's tick ALONE, timed between two events on its stream, from the state the JOINT run is in: its particles are put back afterwards
# and a joint tick follows, so the set keeps evolving as in the G-GPU run (a rank that ticks alone for long drifts away from the frozen
# rows of the others, and its run lists with it: 10 ticks alone at G = 2 took 4.3 ms each against 0.8 in step with the others).
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    ts = []
    for k in range(reps):
        joint_tick()
        keep = r0.theta_all.clone()
        r0.sync()
        e0.record()
        for it in range(c4["n_iters"]):
            r0.local_score(st, None, params[it])
            r0.apply_phi()
        r0.forward_local()
        r0.forward_finish(False)
        e1.record()
        r0.sync()
        ts.append(e0.elapsed_time(e1) * 1e3)
        r0.theta_all.copy_(keep)
    print("cfg4 G=%d n_local=%d aged %d ticks: rank 0's tick alone (GPU time between two events, %d ticks in step with the joint run): median %.1f us, min %.1f, max %.1f" %
          (G, c4["N"] // G, age + rep * reps, reps, float(np.median(ts)), min(ts), max(ts)), flush=True)
