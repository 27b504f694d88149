"""Two (or more) PROCESSES ticking the product shape on ONE GPU at the same time.  The one-launch kernels spin on their own grid; the
hardware scheduler time-slices the processes' queues with wave save / restore, and a partially restored grid deadlocks against the
other process's - a start barrier cannot prevent that.  What the library guarantees: the wait gives up after 50 ms; the owner-computes
kernel commits nothing in that case (tick2.hpp COMMIT) and the library replays the tick on plain kernels, so the caller loses NO tick
(`tick_stats()['replayed']` counts it); from then on the context runs plain kernels only, at the launch-per-iteration rate.  A
time-out of one of the other one-launch forms is reported (DUST_ERR_HIP) - at most one lost tick per context; the child below
re-seeds in that case and reports it.  (The parent never touches the GPU; each child is an ordinary process.)
    python tools/two_process_ticks.py [ticks] [processes]"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dust_amd import Context
T = int(sys.argv[1]); tag = sys.argv[2]
N, S, H = 1024, 128, 30
rng = np.random.default_rng(int(tag))
mu = rng.standard_normal((N, H, 1)).astype(np.float32)
th = (mu + 2 * rng.standard_normal((N, H, 1))).astype(np.float32)
c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=int(tag))
c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
st = np.array([3.0, 0.0], np.float32)
t0 = time.perf_counter()
failed = []
for t in range(T):
    try:
        a_seq, pw = c.svmpc_tick(st, 5)
    except Exception as e:  # a lost tick: its results are invalid and the particles may be partly updated - the controller re-seeds them
        failed.append((t, round(time.perf_counter() - t0, 3), str(e)[:90]))
        assert len(failed) <= 2, failed  # (afterwards the context runs plain kernels: no further time-out can occur)
        c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
        continue
    assert np.isfinite(a_seq).all() and abs(float(pw.sum()) - 1.0) < 1e-3, t
el = time.perf_counter() - t0
print("process %s: %d ticks, %.1f us/tick, paths %s, lost ticks %s" % (tag, T, 1e6 * el / T, c.tick_stats(), failed), flush=True)
'''

if __name__ == "__main__":
    T = sys.argv[1] if len(sys.argv) > 1 else "3000"
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, T, str(i)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(P)]
    rc = 0
    for p in procs:
        out, _ = p.communicate()
        print("\n".join(l for l in out.splitlines() if "amdgpu.ids" not in l))
        rc |= p.returncode
    sys.exit(rc)
