"""A/B timing of library builds on one box: python tools/ab_time.py libA.so libB.so ...  (each in its own process, 3 rounds)."""
import os, subprocess, sys
code = r'''
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from dust_amd import Context
w = bench.WORKLOAD
mu, theta = bench.synth(w["N"], w["H"], 1)
ctx = Context(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"], sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=0, seed=1234)
ctx.set_theta(theta); ctx.set_prior(mu); ctx.set_a_mat(theta)
st = np.array([3.0, 0.0], np.float32)
for _ in range(3000): ctx.svmpc_tick(st, 5, want_outputs=False)
ctx.sync()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(1000): ctx.svmpc_tick(st, 5, want_outputs=False)
    ctx.sync()
    best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
print("%.2f" % best)
'''
libs = sys.argv[1:]
for rnd in range(int(os.environ.get('AB_ROUNDS', '3'))):
    row = []
    for lib in libs:
        env = dict(os.environ, DUST_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        row.append(out.stdout.strip().split("\n")[-1] if out.returncode == 0 else "ERR " + out.stderr[-200:])
    print("round %d: " % rnd + "  ".join("%s=%s" % (os.path.basename(l), v) for l, v in zip(libs, row)), flush=True)
