#!/usr/bin/env python3
"""bench.py - MPC control steps/sec of the SVGD-MPC inner loop on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Workload at every N = BASELINE.json configs[1]: Pendulum, 1024 Stein particles PER GPU, S=128 action samples, M=1,
H=30, 5 SVGD iterations per tick, gpytorch-RBF ("K1") kernel, SGD lr 2, fp32, synthetic seeded inputs resident in HBM.
One "step" = one control tick = SVMPC.optimize(5 iterations) + SVMPC.forward (weights, argmax, roll, prior refresh).
Policy noise is drawn on the device inside the timed region (Philox, fused into the rollout kernel) - the reference also
draws its noise inside the tick - so no work is skipped.

N>1 shards the particle index over the ranks (weak scaling: 1024 particles per GPU, N*1024 in the joint problem) with two
in-place RCCL all-gathers per SVGD iteration (score before the Stein pass, theta after the update - the prior means alias
theta); `value` counts 1024-particle shard-ticks per second summed over ranks (= joint ticks/s * N),
`joint_ticks_per_s` is the joint rate itself.  The pairwise passes are N_loc x N, so per-rank work grows with N by design.

Extra objects in the JSON line (tier contract): `roofline` for the rollout kernel in its HBM-streaming form (external
noise read from HBM, the variant the parity tests drive; run as dust_likelihood_sample runs it - rollouts, costs, weights,
likelihood score, MPPI side update - without the combine of the prior partials, which SURVEY 8d's B_roll does not count), timed with HIP events inside this process (one event pair around
400 back-to-back launches on the context's stream; `traffic` from the committed PMC summary under profiles/); `cpu_baseline` =
the CPU oracle (oracle/dust_oracle.c, a port - the reference is Python and cannot travel) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(model="pendulum", N=1024, S=128, M=1, H=30, n_iters=5, kernel="K1", lr=2.0, alpha=1.0, sigma_a=2.0, sigma_p=2.0)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured achievable)


def synth(N, H, da, seed=0):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + 2.0 * rng.standard_normal((N, H, da))).astype(np.float32)
    return mu, theta


def cpu_baseline(budget_s=12.0):
    """Oracle (C port, OpenMP) on the host cores, same workload, bounded to ~budget_s seconds."""
    from oracle import Oracle, num_threads

    w = WORKLOAD
    o = Oracle(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"])
    mu, theta = synth(w["N"], w["H"], 1)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((w["n_iters"], w["S"], w["N"], w["H"], 1)).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    mix = np.ones(w["N"], np.float32)
    t0 = time.perf_counter()
    ticks = 0
    while True:
        r = o.tick_k1(state, theta, mu, mix, w["sigma_p"], w["sigma_a"], eps, w["n_iters"], w["alpha"], w["lr"], theta)
        theta, mu, mix = r["theta"], r["mu"], r["mix"]
        ticks += 1
        el = time.perf_counter() - t0
        if el > budget_s or ticks >= 200:
            break
    return dict(value=ticks / el, unit="control steps/s", cores=num_threads(), kind="port",
                sample="%d full ticks of the same workload (N=%d,S=%d,H=%d,%d iters) in %.1f s, OpenMP over all host threads"
                       % (ticks, w["N"], w["S"], w["H"], w["n_iters"], el))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--f16-variant", action="store_true", help="also time the rollout kernel over binary16-stored noise (informational)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if world > 1 or (under_launcher and os.environ.get("DUST_BENCH_FORCE_DIST")):
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from dust_amd import Context
    from dust_amd.parallel import ShardedSVMPC

    w = WORKLOAD
    n_loc = w["N"]
    n_tot = n_loc * n_gpus
    mu, theta = synth(n_tot, w["H"], 1)
    state = np.array([3.0, 0.0], np.float32)
    common = dict(model=w["model"], N=n_tot, S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"],
                  sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=local, seed=1234)
    if dist is None:
        ctx = Context(**common)
        ctx.set_theta(theta)
        ctx.set_prior(mu)
        ctx.set_a_mat(theta)

        def tick():
            ctx.svmpc_tick(state, w["n_iters"], want_outputs=False)

        def sync():
            ctx.sync()
    else:
        sh = ShardedSVMPC(common, rank, n_gpus, dist)
        sh.set_state(theta, mu)
        ctx = sh.ctx

        def tick():
            sh.tick(state, w["n_iters"])

        def sync():
            sh.sync()

    for _ in range(args.warmup):
        tick()
    sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tick()
    sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- per-kernel HIP-event timing of the product tick + roofline of the HBM-streaming rollout kernel (rank 0, N=1 form)
    per_kernel, roofline = {}, None
    if rank == 0 and not args.no_roofline:
        c1 = Context(**dict(common, N=n_loc))
        mu1, th1 = synth(n_loc, w["H"], 1)
        c1.set_theta(th1)
        c1.set_prior(mu1)
        c1.set_a_mat(th1)
        c1.profile(True)
        for _ in range(20):
            c1.svmpc_tick(state, w["n_iters"], want_outputs=False)
        c1.sync()
        per_kernel = {k: dict(avg_us=1e3 * ms / n, launches=n) for k, (ms, n) in c1.profile_get().items()}
        # HBM-streaming form: eps [iters][S][N][D] resident in HBM, one distinct slice per launch
        n_slices = 8
        slice_f = w["S"] * n_loc * w["H"]
        ptr = c1.device_noise(n_slices * slice_f, seed=99)
        avg_s = c1.profile_rollout(state, ptr, n_slices, 400) * 1e-3
        bytes_alg = c1.rollout_bytes()
        ach = bytes_alg / avg_s / 1e9
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "round1_rollout_traffic.json")
        if os.path.exists(tf):  # PMC passes cannot run inside this process: the committed summary of the same kernel / shape
            with open(tf) as fh:
                tj = json.load(fh)
            traffic, traffic_src = tj["hbm_bytes_per_launch"], "profiles/round1_rollout_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        roofline = dict(kernel="dust::rollout_stream_kernel<0> (rollout kernel, HBM-streaming form: caller-supplied eps)", bound="hbm", achieved=ach, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                        algorithmic_bytes_per_launch=bytes_alg, avg_launch_us=avg_s * 1e6, launches=400,
                        timing="one HIP event pair around 400 back-to-back launches on the context's stream")
        c1.device_free(ptr)
        if args.f16_variant:
            # the same kernel over binary16-stored noise (BASELINE.json config 5 "fp16 rollout"): informational, never the roofline
            # line; opt-in, so that the default run's rocprofv3 row of this kernel holds the fp32 launches only
            ptr16 = c1.device_noise(n_slices * slice_f, seed=99, f16=True)
            t16 = c1.profile_rollout(state, ptr16, n_slices, 400, f16=True) * 1e-3
            b16 = c1.rollout_bytes(eps_f16=True)
            roofline["f16_storage_variant"] = dict(avg_launch_us=t16 * 1e6, algorithmic_bytes_per_launch=b16, achieved=b16 / t16 / 1e9,
                                                   frac=b16 / t16 / 1e9 / HBM_PEAK_GBS)
            c1.device_free(ptr16)
        c1.profile(False)
        c1.close()

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        joint = args.steps / el
        out = {
            "metric": "MPC control steps/sec (SVGD-MPC tick: 5 SVGD iterations + forward)",
            "value": joint * n_gpus,
            "unit": "control steps/s (1024-particle shard-ticks summed over GPUs)",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Pendulum N=%d particles/GPU (%d total), S=128, M=1, H=30, 5 SVGD iters, K1 (gpytorch-RBF) kernel, SGD, "
                                   "device Philox noise inside the tick" % (n_loc, n_tot),
                       "parallelism": "particles sharded x%d, two in-place RCCL all-gathers (score, theta) per SVGD iteration" % n_gpus if n_gpus > 1 else "single GPU"},
            "joint_ticks_per_s": joint,
            "n_particles_total": n_tot,
            "per_kernel": per_kernel,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
