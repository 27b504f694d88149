"""Why is the driver's short run (bench.py --steps 20 --warmup 5) slower per tick than a 300-step run?  Times the open-loop cfg2
tick in bursts of 20 after warm-ups of different length and after idle gaps (clock ramp vs. one-off runtime costs)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
if os.environ.get("WITH_TORCH", "1") == "1":
    import torch
    torch.cuda.synchronize()
from dust_amd import Context
import bench

w = bench.WORKLOAD
mu, theta = bench.synth(w["N"], w["H"], 1)
ctx = Context(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"],
              sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=0, seed=1234)
ctx.set_theta(theta); ctx.set_prior(mu); ctx.set_a_mat(theta)
st = np.array([3.0, 0.0], np.float32)


def burst(n):
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.svmpc_tick(st, w["n_iters"], want_outputs=False)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6


print("first bursts after context creation (us/tick incl. final sync | enqueue only):")
for i in range(8):
    a, b = burst(5 if i == 0 else 20)
    print("  burst %d: %.1f | %.1f" % (i, a, b))
for gap in (0.0, 0.001, 0.01, 0.1, 1.0):
    for wu in (0, 5, 50, 400):
        time.sleep(gap)
        if wu:
            burst(wu)
        a, b = burst(20)
        print("idle %.3f s, warm-up %3d ticks: 20-tick burst %.1f us/tick (enqueue %.1f)" % (gap, wu, a, b))
for n in (20, 50, 100, 300, 1000):
    a, b = burst(n)
    print("burst of %4d: %.1f us/tick (enqueue %.1f)" % (n, a, b))
