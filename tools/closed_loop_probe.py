"""Where a closed-loop tick's time goes: kernel alone (open loop), + one stream synchronisation per tick, + outputs (device-to-host copy),
+ host plant.  python tools/closed_loop_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd import Context

w = bench.WORKLOAD
mu, theta = bench.synth(w["N"], w["H"], 1)
ctx = Context(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"], sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=0, seed=1234)
ctx.set_theta(theta); ctx.set_prior(mu); ctx.set_a_mat(theta)
st = np.array([3.0, 0.0], np.float32)
for _ in range(4000): ctx.svmpc_tick(st, 5, want_outputs=False)
ctx.sync()
n = 2000
def run(name, body):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): body()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    print("%-58s %7.2f us per tick" % (name, best), flush=True)
run("open loop (enqueue only)", lambda: ctx.svmpc_tick(st, 5, want_outputs=False))
def a():
    ctx.svmpc_tick(st, 5, want_outputs=False); ctx.sync()
run("+ stream synchronisation per tick", a)
run("+ outputs (a_seq, p_weights)", lambda: ctx.svmpc_tick(st, 5, want_outputs=True))
state = [st.copy()]
def c():
    a_seq, _ = ctx.svmpc_tick(state[0], 5, want_outputs=True)
    state[0] = bench.pendulum_plant(state[0], a_seq[0, 0])
run("+ host plant (bench.py's closed loop)", c)
t0 = time.perf_counter()
for _ in range(20000): bench.pendulum_plant(st, 0.3)
print("host plant alone %.2f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
# ---- round 5: closed-loop serving (outputs through pinned memory + the next tick launched ahead of its state)
for wait in (0.0, 2000.0):
    ctx.serve_start(5, wait)
    tag = "served, wait_us=%g" % wait
    run(tag + ": outputs only (constant state)", lambda: ctx.svmpc_tick(st, 5, want_outputs=True))
    run(tag + ": + host plant", c)
    def d():
        a_seq, _ = ctx.svmpc_tick(state[0], 5, want_outputs="action")
        state[0] = bench.pendulum_plant(state[0], a_seq[0, 0])
    run(tag + ": a_seq only + host plant", d)
    print("   tick paths:", ctx.tick_stats(), flush=True)
    ctx.serve_stop()
