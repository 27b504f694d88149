"""Per-iteration GPU time of ONE rank's share of the weak-scaled bench (1024 particles per GPU, N = 1024 * G in the joint
problem), measured on a single GPU without collectives: what each rank computes between the all-gathers."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dust_amd.parallel import DeviceShard

for G in (() if (len(sys.argv) > 1 and sys.argv[1] == "cfg4") else (1, 2, 4, 8)):
    N = 1024 * G
    cfg = dict(model="pendulum", N=N, S=128, M=1, H=30, kernel="K1", lr=2.0, sigma_a=2.0, sigma_p=2.0, seed=3)
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, 30, 1)).astype(np.float32)
    th = (mu + 2 * rng.standard_normal((N, 30, 1))).astype(np.float32)
    sh = DeviceShard(cfg, 0, G, use_torch_stream=False)
    sh.set_state(th, mu)
    state = np.array([3.0, 0.0], np.float32)
    for rep in range(2):
        sh.ctx.sync()
        t0 = time.perf_counter()
        iters = 200
        for _ in range(iters):
            sh.local_score(state)
            sh.apply_phi()
        sh.ctx.sync()
        el = (time.perf_counter() - t0) / iters
    sh.ctx.profile(True)
    for _ in range(20):
        sh.local_score(state)
        sh.apply_phi()
    sh.ctx.sync()
    pk = {k: round(1e3 * ms / n, 1) for k, (ms, n) in sh.ctx.profile_get().items()}
    print("G=%d N=%d: %.1f us per iteration (local score + Stein/update), unfused kernels: %s" % (G, N, el * 1e6, pk), flush=True)
    sh.ctx.close()

# the strong-scaled bench (`bench.py --gpus G`: BASELINE cfg4, Particle N = 16384 in total): one rank's share, collectives excluded
if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
    from bench import particle_grid

    for G in ([int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
        N, S, M, H = 16384, 64, 4, 40
        cfg = dict(model="particle", N=N, S=S, M=M, H=H, kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0, uncertain_params=("mass",),
                   grid=particle_grid(), seed=3)
        rng = np.random.default_rng(0)
        mu = rng.standard_normal((N, H, 2)).astype(np.float32)
        th = (mu + rng.standard_normal((N, H, 2))).astype(np.float32)
        sh = DeviceShard(cfg, 0, G, use_torch_stream=False)
        sh.set_state(th, th)
        # the prior means alias theta, as after every forward(): one rank-local forward (a sharded context refreshes its prior there)
        sh.local_score(np.array([-9.0, -9.0, 0.0, 0.0], np.float32), params=(1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32))
        sh.apply_phi()
        sh.forward_local()
        sh.forward_finish()
        state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
        params = (1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
        for rep in range(2):
            sh.ctx.sync()
            t0 = time.perf_counter()
            iters = 20
            for _ in range(iters):
                sh.local_score(state, params=params)
                sh.apply_phi()
            sh.ctx.sync()
            el = (time.perf_counter() - t0) / iters
        sh.ctx.profile(True)
        for _ in range(5):
            sh.local_score(state, params=params)
            sh.apply_phi()
        sh.ctx.sync()
        pk = {k: round(1e3 * ms / n, 1) for k, (ms, n) in sh.ctx.profile_get().items()}
        print("cfg4 G=%d n_local=%d: %.0f us per iteration (local score + Stein/update), kernels: %s" % (G, N // G, el * 1e6, pk), flush=True)
        sh.ctx.profile(False)
        # The whole tick of one rank without its collectives: iteration + forward (local log p, finalize over all N, roll).  The MEDIAN of three
        # repetitions (the particle set evolves from repetition to repetition, and with it the share of exactly-zero kernel blocks: the
        # minimum would pick the sparsest set) with the collector off: round 3's profile (and this round's first pass) showed 2 690 us at 8 shards against 630 us in
        # every other run - in the kernel trace ONE 44 ms gap on the host side between a roll_kernel and the next set_state_kernel
        # (the fourth context of the process: the three before it had just released ~3 GB of device and host buffers), i.e. 2.2 ms
        # when averaged over the 20 ticks of the timed loop; every kernel of those ticks ran at its usual time.
        gc.collect()
        gc.disable()
        reps2 = []
        for rep in range(3):
            sh.ctx.sync()
            t0 = time.perf_counter()
            for _ in range(iters):
                sh.local_score(state, params=params)
                sh.apply_phi()
                sh.forward_local()
                sh.forward_finish()
            sh.ctx.sync()
            reps2.append((time.perf_counter() - t0) / iters)
        gc.enable()
        el2 = sorted(reps2)[1]
        print("        whole tick of the rank (collectives excluded): %.0f us -> forward %.0f us" % (el2 * 1e6, (el2 - el) * 1e6), flush=True)
        sh.ctx.profile(True)  # per-kernel times of the whole tick (profiling mode: one event pair per launch)
        for _ in range(5):
            sh.local_score(state, params=params)
            sh.apply_phi()
            sh.forward_local()
            sh.forward_finish()
        sh.ctx.sync()
        print("        per kernel, whole tick: %s" % {k: round(1e3 * ms / n, 1) for k, (ms, n) in sh.ctx.profile_get().items()}, flush=True)
        sh.ctx.close()
