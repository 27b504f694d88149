#!/usr/bin/env python3
"""bench.py - MPC control steps/sec of the SVGD-MPC inner loop on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

N = 1 (the contract's workload): BASELINE.json configs[1] - Pendulum, 1024 Stein particles, S=128 action samples, M=1, H=30,
5 SVGD iterations per tick, gpytorch-RBF ("K1") kernel, SGD lr 2, fp32, synthetic seeded inputs resident in HBM.  One "step" =
one control tick = SVMPC.optimize(5 iterations) + SVMPC.forward (weights, argmax, roll, prior refresh): ONE persistent kernel
launch (dust_amd/csrc/tick2.hpp; the first tick of a context, whose prior does not alias the particles yet, runs plain kernels).  Policy noise is drawn on the device inside the timed region (Philox, in registers) - the
reference also draws its noise inside the tick - so no work is skipped.  `value` is the open-loop rate (ticks enqueued back to
back, plant state constant); `closed_loop_ticks_per_s` is the rate of the loop of dust/utils/simulations.py:104-123 - every tick's
chosen sequence comes back to the host, its first action steps a host plant model, the new state feeds the next tick - run through
closed-loop serving (dust_svmpc_serve_start: outputs through pinned host memory, the next tick launched ahead of its state);
`closed_loop` carries the unserved figure of the same loop (one device-to-host copy + one stream synchronisation per tick) beside it.

N > 1: BASELINE.json configs[3] - Particle (2-D point mass, obstacle grid), 16384 Stein particles, S=64, M=4, H=40, 1 SVGD
iteration per tick - sharded over the N ranks by particle index (strong scaling: the joint problem is fixed), with the in-place
RCCL all-gathers of DESIGN.md section 6; `value` = joint ticks/s.  `weak_cfg2` carries the round-1 weak-scaled figure
(1024 Pendulum particles per GPU) as a secondary field when --weak is given.

Warm-up: the W warm-up ticks are run, and then more of them until 1 s has passed (`warmup_ticks_run` says how many): the
chip's clock needs load to settle - after the idle seconds of process start-up, 20 timed ticks take 137 us each behind 0.1 s of
ticks and 126 us behind 0.3 s or more, whatever the kernel (tools/coldstart.py, tools/bench_timing_probe.py) - and the driver's
W = 5 ticks are 0.6 ms.  The K timed ticks are exactly K (garbage collection off between the two clock readings).  torch's HIP context is created before the warm-up (its lazy creation
inside the first torch.cuda.synchronize() stalled the 20 ticks behind it for 37 ms).

`scale_workload` (every line, N = 1 included): BASELINE.json configs[3] (Particle N=16384, S=64, M=4, H=40, 1 iteration) on the
N GPUs of the run - ticks/s, ms per tick and, for N > 1, the tick's all-gathers timed alone (comm_us_per_tick) - so that
value(N) / value(1) can be formed on ONE workload from the driver's lines: scale_workload.ticks_per_s.

Extra objects in the JSON line (tier contract):
  roofline      top level: the TIMED kernel (svmpc_tick2_kernel, one launch = one control tick): SURVEY 8d algorithmic bytes over the
                measured tick (`achieved` GB/s, `hbm_frac`) and what actually bounds it - VALU issue and hand-off latency (`frac` =
                issue floor / measured, from the committed SQ_INSTS_VALU pass).  `rollout_kernel`: the rollout kernel in its HBM-BOUND
                form - stored states (MultiDISCO.forward returns them), BASELINE configs[2] size: Particle N=4096, S=64, M=64, H=40 ->
                11.1 GB written per launch, far beyond the 256 MiB Infinity Cache - timed with HIP events on the context's stream;
                `forms` adds the other rollout forms.  `traffic` figures come from committed PMC summaries under profiles/ (PMC
                passes cannot run inside this process); each summary carries a stamp of the kernel sources it was measured on
                (tools/srcstamp.py) and is reported as null when the sources have changed since.
  cpu_baseline  the CPU oracle (oracle/dust_oracle.c, a scalar C port - the reference is Python and cannot travel) on the host:
                all threads on whole ticks, and ONE core on one SVGD iteration of the same tick; CPU model string included.
                It is a checker, not a tuned CPU implementation: no credit attaches to the ratio.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(model="pendulum", N=1024, S=128, M=1, H=30, n_iters=5, kernel="K1", lr=2.0, alpha=1.0, sigma_a=2.0, sigma_p=2.0)
CFG3 = dict(model="particle", N=4096, S=64, M=64, H=40)            # roofline: stored-states form
CFG4 = dict(model="particle", N=16384, S=64, M=4, H=40, n_iters=1)  # multi-GPU workload
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured achievable)
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2  # 1024 SIMD-32 units, one wave64 VALU instruction per 2 cycles at 2.4 GHz


def far_shares(ctx):
    """Share of the pairwise units the last tick's two large-set passes left out (dust_debug_far_units / dust_debug_far_logp: debug
    entries of the library, not part of include/dust_amd.h); None where a pass ran without the pre-pass."""
    import ctypes as C
    from dust_amd import _lib as L
    lib = L.load()
    out = {}
    for key, fn in (("fused_prior_stein_pass", "dust_debug_far_units"), ("log_p_pass", "dust_debug_far_logp")):
        f = getattr(lib, fn)
        f.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        f.restype = C.c_int
        v = (C.c_longlong * 2)()
        out[key] = (v[0] / v[1]) if (f(ctx._h, v) == 0 and v[1]) else None
    # the run lists of pairwise_packed.hpp: packed units walked by the last pass 1 over the (query tile, key chunk) units of the set
    IP = C.POINTER(C.c_int)
    lib.dust_debug_pack_lists.argtypes = [C.c_void_p, C.c_int, IP, IP, IP, IP, C.POINTER(C.c_uint), IP]
    lib.dust_debug_pack_lists.restype = C.c_int
    nu, al = C.c_int(), C.c_int()
    if lib.dust_debug_pack_lists(ctx._h, -1, C.byref(nu), C.byref(al), None, None, None, None) == 0 and al.value:
        out["run_list_units_over_all_units"] = nu.value / al.value
    return out


def synth(N, H, da, seed=0, spread=2.0):
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + spread * rng.standard_normal((N, H, da))).astype(np.float32)
    return mu, theta


def particle_grid():
    """220 x 220 cells of 0.1 m, 4 x 4 block obstacles (the demo's grid_4x4 pattern; synthetic occupancy)."""
    g = np.zeros((220, 220), np.float32)
    for bx in range(4):
        for by in range(4):
            x0, y0 = 30 + bx * 45, 30 + by * 45
            g[x0:x0 + 18, y0:y0 + 18] = 1.0
    return g


def pendulum_plant(state, u, dt=0.05, g=9.8, m=1.0, l=1.0):
    """Host plant step for the closed-loop figure (the stand-in plant of dust_amd/utils/simulations.py: PendulumModel.step,
    pendulum.py:61-100, nominal parameters)."""
    th, thd = float(state[0]), float(state[1])
    u = min(max(float(u), -2.0), 2.0)
    thd = thd + dt * (-3.0 * g / (2.0 * l) * np.sin(th + np.pi) + 3.0 * u / (m * l * l))
    thd = min(max(thd, -8.0), 8.0)
    return np.array([th + thd * dt, thd], np.float32)


def profile_json(name, family=None):
    """Newest committed summary profiles/round<k>_<name>.json (k = 6, 5, 4); with `family`, only if its source stamp matches the kernel
    sources of this checkout (tools/srcstamp.py) - a summary measured on other sources is not this kernel's: (None, why)."""
    for rnd in (6, 5, 4):
        pth = os.path.join(ROOT, "profiles", "round%d_%s.json" % (rnd, name))
        if not os.path.exists(pth):
            continue
        with open(pth) as fh:
            j = json.load(fh)
        rel = "profiles/round%d_%s.json" % (rnd, name)
        if family is None:
            return j, rel
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import srcstamp

        if j.get("source_stamp") == srcstamp.stamp(family):
            return j, rel
        return None, "%s was measured on other kernel sources (stamp %s, now %s): traffic not reported" % (rel, j.get("source_stamp"), srcstamp.stamp(family))
    return None, "no committed summary"


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores(n_omp):
    """Host cores this process may actually use: the affinity mask and the cgroup CPU quota, capped by OpenMP's own count."""
    n = n_omp
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, int(q / int(fh2.read().split()[0]) + 0.5)))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(budget_s=10.0):
    """Oracle (scalar C port, OpenMP) on the host cores, same workload; bounded to ~budget_s per leg."""
    from oracle import Oracle, num_threads, set_num_threads

    w = WORKLOAD
    o = Oracle(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"])
    mu, theta = synth(w["N"], w["H"], 1)
    rng = np.random.default_rng(1)
    eps = rng.standard_normal((w["n_iters"], w["S"], w["N"], w["H"], 1)).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    mix = np.ones(w["N"], np.float32)
    n_omp = num_threads()
    n_all = usable_cores(n_omp)  # (OpenMP counts the machine's hardware threads; a container's CPU quota / affinity mask can be far smaller -
    set_num_threads(n_all)       #  round 3 ran 128 threads on a quota of a few cores: 1.4x the one-core rate)
    t0 = time.perf_counter()
    ticks = 0
    th, m_, mx = theta, mu, mix
    while True:
        r = o.tick_k1(state, th, m_, mx, w["sigma_p"], w["sigma_a"], eps, w["n_iters"], w["alpha"], w["lr"], th)
        th, m_, mx = r["theta"], r["mu"], r["mix"]
        ticks += 1
        el = time.perf_counter() - t0
        if el > budget_s or ticks >= 200:
            break
    all_rate = ticks / el
    # ONE core: one SVGD iteration of the same tick (a whole tick on one core is minutes), scaled by the iteration count
    one = None
    if set_num_threads(1):
        t1 = time.perf_counter()
        o.tick_k1(state, theta, mu, mix, w["sigma_p"], w["sigma_a"], eps[:1], 1, w["alpha"], w["lr"], theta)
        e1 = time.perf_counter() - t1
        set_num_threads(n_omp)
        one = dict(value=1.0 / (e1 * w["n_iters"]), seconds_per_iteration=e1,
                   sample="1 SVGD iteration (+ forward) of the same tick on 1 thread, scaled by the 5 iterations of a tick")
    return dict(value=all_rate, unit="control steps/s", cores=n_all, kind="port", cpu=cpu_model(), hardware_threads=n_omp,
                sample="%d full ticks of the same workload (N=%d,S=%d,H=%d,%d iters) in %.1f s, OpenMP over the %d host threads this process may use "
                       "(affinity mask / cgroup quota; the machine has %d)" % (ticks, w["N"], w["S"], w["H"], w["n_iters"], el, n_all, n_omp),
                one_core=one,
                note="scalar C restatement used as the parity checker (O(S N^2 D) work of the reference skipped where it is discarded); "
                     "not a tuned CPU implementation - no credit attaches to the GPU/CPU ratio")


def roofline_section(local, state_pend):
    """Roofline of the rollout kernel (HIP events inside this process) + the product kernel's issue fraction."""
    from dust_amd import Context

    out = {}
    # ---- (1) HBM-bound form: stored states at cfg3 size
    c3 = CFG3
    rng = np.random.default_rng(3)
    th = rng.standard_normal((c3["N"], c3["H"], 2)).astype(np.float32)
    params = (1.0 + 0.1 * rng.standard_normal((c3["M"], 1))).astype(np.float32)
    c = Context(model="particle", N=c3["N"], S=c3["S"], M=c3["M"], H=c3["H"], kernel="K1", sigma_a=1.0, sigma_p=1.0,
                uncertain_params=("mass",), grid=particle_grid(), seed=7, device=local)
    c.set_theta(th); c.set_prior(th); c.set_a_mat(th)
    st = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    n_slices = 4
    slice_f = c3["S"] * c3["N"] * c3["H"] * 2
    ptr = c.device_noise(n_slices * slice_f, seed=99)
    eps0 = rng.standard_normal((c3["S"], c3["N"], c3["H"], 2)).astype(np.float32)
    c.likelihood_sample(st, eps0, params)  # uploads the dynamics samples (they stay on the device)
    reps = 20
    both_s = c.profile_rollout(st, ptr, n_slices, reps, store_states=True) * 1e-3  # back-to-back launches, one event pair
    per_kernel = c.profile_get()  # ... and one event pair per launch, per kernel (dust_profile_rollout leaves them in the slots)
    samples = []  # single launches, one event pair each: the spread (launches of this kernel range 2.0-2.8 ms run to run)
    for _ in range(16):
        c.profile_rollout(st, ptr, n_slices, 1, store_states=True)
        pk = c.profile_get()
        if "states_kernel" in pk:
            samples.append(pk["states_kernel"][0] / pk["states_kernel"][1] * 1e3)
    b_alg = c.rollout_bytes(store_states=True)
    kname = "dust::rollout_stream_kernel<1,true,false,true> (rollout kernel, stored-states form: states [M][S][N][H+1][ds] written)"
    avg_s, second_pass = both_s, None
    if "states_kernel" in per_kernel:
        # the whole-line form: rollouts + states + costs in particle_states_kernel, then the regular kernel's injected-costs pass
        ms_k, n_k = per_kernel["states_kernel"]
        avg_s = ms_k / n_k * 1e-3
        ms_2, n_2 = per_kernel["rollout_kernel"]
        second_pass = ms_2 / n_2 * 1e3
        kname = ("dust::particle_states_kernel<2> (stored-states rollouts, whole-line form: states [M][S][N][H+1][ds] written; its "
                 "launch pair includes the (idle) general-path launch behind it)")
    ach = b_alg / avg_s / 1e9
    tj, traffic_src = profile_json("rollout_states_traffic", "states")
    traffic = tj["hbm_bytes_per_launch"] if tj else None
    if tj:
        traffic_src += " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; source stamp %s)" % tj["source_stamp"]
    out.update(kernel=kname, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
               algorithmic_bytes_per_launch=b_alg, avg_launch_us=avg_s * 1e6, launches=reps,
               launch_us_median=(float(np.median(samples)) if samples else None),
               launch_us_p90=(float(np.percentile(samples, 90)) if samples else None),
               frac_at_median=(b_alg / (float(np.median(samples)) * 1e-6) / 1e9 / HBM_PEAK_GBS if samples else None),
               workload="Particle N=%d, S=%d, M=%d, H=%d (BASELINE configs[2]): %.2f GB of states per launch (working set >> 256 MiB Infinity Cache)"
                        % (c3["N"], c3["S"], c3["M"], c3["H"], b_alg / 1e9),
               timing=("one HIP event pair per launch on the context's stream, %d launches" % reps) if second_pass is not None
               else ("one HIP event pair around %d back-to-back launches on the context's stream" % reps))
    if second_pass is not None:
        out["second_pass_us"] = second_pass  # softmax / weights / score from the injected costs (rollout_stream_kernel, costs_in mode)
        out["both_passes"] = dict(avg_us=both_s * 1e6, achieved=b_alg / both_s / 1e9, frac=b_alg / both_s / 1e9 / HBM_PEAK_GBS,
                                  timing="one HIP event pair around %d back-to-back (states kernel + second pass) pairs" % reps)
    # the same shape without stored states (compute / latency bound: 86 MB per launch)
    avg_ns = c.profile_rollout(st, ptr, n_slices, reps) * 1e-3
    b_ns = c.rollout_bytes()
    forms = {"cfg3_no_store": dict(avg_launch_us=avg_ns * 1e6, algorithmic_bytes_per_launch=b_ns, achieved=b_ns / avg_ns / 1e9,
                                   frac=b_ns / avg_ns / 1e9 / HBM_PEAK_GBS)}
    c.device_free(ptr)
    c.close()
    # ---- (2) cfg2 no-store form over a noise working set beyond the Infinity Cache (20 x 15.7 MB = 315 MB)
    w = WORKLOAD
    c1 = Context(model="pendulum", N=w["N"], S=w["S"], M=1, H=w["H"], kernel="K1", lr=w["lr"], sigma_a=2.0, sigma_p=2.0, seed=1234, device=local)
    mu1, th1 = synth(w["N"], w["H"], 1)
    c1.set_theta(th1); c1.set_prior(mu1); c1.set_a_mat(th1)
    n2 = 20
    sf = w["S"] * w["N"] * w["H"]
    p2 = c1.device_noise(n2 * sf, seed=98)
    a2 = c1.profile_rollout(state_pend, p2, n2, 400) * 1e-3
    b2 = c1.rollout_bytes()
    forms["cfg2_no_store"] = dict(avg_launch_us=a2 * 1e6, algorithmic_bytes_per_launch=b2, achieved=b2 / a2 / 1e9, frac=b2 / a2 / 1e9 / HBM_PEAK_GBS,
                                  working_set_mb=n2 * sf * 4 / 1e6,
                                  note="131072 rollouts = 2 waves per SIMD: one resident wave of workgroups, latency-bound (DESIGN.md section 5)")
    c1.device_free(p2)
    c1.close()
    # ---- (3) the other stored-states forms, with what BOUNDS them: HBM fraction from tools/states_probe.py and the
    # VALU issue fraction from a rocprofv3 --pmc SQ_INSTS_VALU pass of the same probe (tools/measure_round4.sh -> committed summaries).
    # The Pendulum forms write 8 (fp32) / 4 (binary16) bytes per ~45 vector instructions per lane: they are VALU-issue bound, not HBM bound.
    hb, hsrc = profile_json("states_hbm")
    vj, vsrc = profile_json("states_forms")
    if hb and vj:

        def issue(sub):
            for k, e in vj.items():
                if sub in k:
                    return e["valu_issue_frac"]
            return None

        for key, kern, bound in (("pendulum_store_f32", "pendulum_states_kernel<false>", "valu"),
                                 ("pendulum_store_f16", "pendulum_states_f16_kernel<false>", "valu"),
                                 ("particle_store_f16", "particle_states_f16_kernel", "valu (whole-line kernel of round 4: 160 instructions per step and pair at 2 waves per SIMD)")):
            if key in hb:
                e = hb[key]
                forms[key] = dict(kernel="dust::" + kern, bound=bound, avg_launch_us=e["total_us"], hbm_frac=e["hbm_frac"], achieved_gbs=e["achieved_gbs"],
                                  valu_issue_frac=issue(kern), workload="%s N=%d S=%d M=%d H=%d" % (e["model"], e["N"], e["S"], e["M"], e["H"]),
                                  source="%s, %s (fraction of one wave64 VALU instruction per 2 cycles per SIMD; tools/valu_rate_probe.hip "
                                         "prices the mix at 2.6-8.2 cycles per instruction)" % (hsrc, vsrc))
    out["forms"] = forms
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--open-loop-only", action="store_true",
                    help="N=1: the timed open-loop leg and nothing else on the GPU (no cfg4 leg, no roofline legs, no closed loops, no CPU baseline): "
                         "under rocprofv3 the svmpc_tick2_kernel row is then the timed kernel alone (VERDICT r5 item 3)")
    ap.add_argument("--weak", action="store_true", help="N>1: also time the weak-scaled cfg2 form (1024 Pendulum particles per GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import __graft_entry__ as entry

    # Exactly ONE line on stdout: the JSON.  Libraries in the process write to the C-level stdout as well (RCCL prints its version banner
    # there when NCCL_DEBUG asks for it - it landed BEHIND the JSON line in the world-1 runs of this round): file descriptor 1 is pointed
    # at stderr for the run, and the JSON line goes out through a duplicate of the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if rank == 0:  # (before anything initialises the GPU in this process: a stale library would compile in a child process)
        entry.build()
    # Order of the single-GPU run: the cfg4 leg and the roofline legs (GPU, seconds), THEN the CPU baseline (host, ~10 s),
    # and the timed cfg2 leg and its closed loops LAST - a driver that samples device activity a few times over the run finds the GPU at
    # work at both ends instead of idle behind (r4) or in front of (r5) ten seconds of host arithmetic
    cpu = None
    import torch

    # torch initialises its HIP context lazily, at the first torch.cuda call - which would otherwise be the synchronize() in front of
    # the timed region: the 20 ticks enqueued right behind it then took 37 ms instead of 0.13 (tools/bench_timing_probe.py)
    torch.cuda.set_device(local)
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()

    dist = None
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if world > 1 or (under_launcher and os.environ.get("DUST_BENCH_FORCE_DIST")):
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    n_gpus = max(world, 1)
    if args.gpus != n_gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if dist is not None:
        dist.barrier()
    from dust_amd import Context
    from dust_amd.parallel import ShardedSVMPC

    # clock ramp after the idle seconds of process start-up (and of the CPU baseline leg): 0.1 s of load left the 20 timed ticks at 137 us,
    # 0.3 s and more at 126 us (tools/coldstart.py, tools/bench_timing_probe.py); ONE first run on a fresh box in round 6 still came out 8 %
    # slow behind 0.4 s (profiles/round6_bench_driver_cmd_first_run_on_box.json): 1 s now - warm-up is untimed, >= W ticks as the contract asks
    MIN_WARM_S = float(os.environ.get("DUST_BENCH_WARM_S", "1.0"))

    def warm(tick, sync, n_min, min_s=None):
        """>= n_min warm-up ticks and >= min_s (MIN_WARM_S) seconds of them, enqueued in small batches (the GPU, not the host queue, keeps time)."""
        n, t0 = 0, time.perf_counter()
        min_s = MIN_WARM_S if min_s is None else min_s
        if dist is not None:  # a sharded tick is a collective: every rank must run the same number (cfg4 ticks are 1-4 ms each)
            n_min = max(n_min, 30)
        while n < n_min or (dist is None and time.perf_counter() - t0 < min_s):
            for _ in range(max(1, min(25, n_min - n) if n < n_min else 25)):
                tick()
                n += 1
            sync()
        return n

    def timed(tick, sync):
        n_w = warm(tick, sync, args.warmup)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        gc_was = gc.isenabled()
        gc.disable()  # (a collection inside a 2 ms region is a 5-10 % error; timeit does the same)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tick()
        t_enq = time.perf_counter()
        torch.cuda.synchronize()  # (a device-wide wait: it covers the context's own stream - no separate stream synchronisation in front of it)
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        sync()  # (outside the timed region: surfaces a time-out of one of the ticks as an error; a REPLAYED tick - one whose launch did not
                #  start and was run late by this call - would not have been paid for inside the region: the caller checks `replayed`)
        if os.environ.get("DUST_BENCH_DEBUG"):
            print("timed region: enqueue %.1f us, total %.1f us, warm-up ticks %d" % ((t_enq - t0) * 1e6, el * 1e6, n_w), file=sys.stderr)
        if dist is not None:
            t = torch.tensor([el], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, n_w

    w = WORKLOAD
    state = np.array([3.0, 0.0], np.float32)
    extra = {}
    rk_early = None
    if args.open_loop_only:
        args.no_roofline = args.no_cpu_baseline = True
    if dist is None:
        # the multi-GPU workload on this one GPU: the N = 1 point of the scaling curve
        if not args.open_loop_only:
            c4 = CFG4
            mu4, theta4 = synth(c4["N"], c4["H"], 2, spread=1.0)
            one = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                          uncertain_params=("mass",), grid=particle_grid(), device=local, seed=1234)
            one.set_theta(theta4); one.set_prior(mu4); one.set_a_mat(theta4)
            p4 = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
            st4 = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
            # (0.4 s: the AGE of the particle set the cfg4 figures of rounds 5-6 are quoted at - ~300-400 ticks; the set, and with it every
            #  data-dependent shortcut of the large-set passes, evolves with the ticks: 0.95-1.00 ms there, 1.04-1.06 ms behind 1 s of ticks)
            n4w = warm(lambda: one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False), one.sync, 3, min_s=0.4)
            n4 = 20
            t1 = time.perf_counter()
            for _ in range(n4):
                one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False)
            one.sync()
            e4 = time.perf_counter() - t1
            far4 = far_shares(one)
            one.close()
            # the same ticks with every (query tile, key chunk) unit of the pairwise passes visited (DUST_FAR=0: pairwise_far.hpp off) -
            # what a particle set WITHOUT far pairs costs (the pre-pass then finds nothing to leave out)
            os.environ["DUST_FAR"] = "0"
            try:
                allv = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                               uncertain_params=("mass",), grid=particle_grid(), device=local, seed=1234)
            finally:
                os.environ.pop("DUST_FAR", None)
            allv.set_theta(theta4); allv.set_prior(mu4); allv.set_a_mat(theta4)
            for _ in range(n4w + n4 - 10):  # (the same age as the timed ticks above: the set, and with it every data-dependent shortcut, evolves)
                allv.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False)
            allv.sync()
            t1 = time.perf_counter()
            for _ in range(10):
                allv.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False)
            allv.sync()
            e4a = (time.perf_counter() - t1) / 10
            allv.close()
            extra["scale_workload"] = dict(workload="Particle N=16384, S=64, M=4, H=40, 1 SVGD iter, K1, SGD, device Philox noise (BASELINE configs[3])",
                                           n_gpus=1, ticks_per_s=n4 / e4, ms_per_tick=1e3 * e4 / n4, comm_us_per_tick=0.0, ticks=n4,
                                           warmup_ticks_run=n4w, pairwise_units_left_out=far4, ms_per_tick_all_units_visited=1e3 * e4a,
                                           note="pairwise_units_left_out: share of the (query tile, key chunk) units of the fused prior / Stein "
                                                "pass and of the log-p pass whose every term is below 2^-43 of its sum's leading term "
                                                "(pairwise_far.hpp: a binary16 MFMA bound decides, the result moves by < half an ulp of that "
                                                "term); run_list_units_over_all_units: 64-key units pass 1 actually walks (the near keys of a "
                                                "query tile packed, tiles in leader order: pairwise_packed.hpp) over the units of the set. "
                                                "DATA DEPENDENT: a clustered set has no far pair and runs at ms_per_tick_all_units_visited")
        if rank == 0 and not args.no_roofline:
            rk_early = roofline_section(local, state)
        if rank == 0 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        mu, theta = synth(w["N"], w["H"], 1)
        ctx = Context(model=w["model"], N=w["N"], S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"],
                      sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=local, seed=1234)
        ctx.set_theta(theta)
        ctx.set_prior(mu)
        ctx.set_a_mat(theta)
        for attempt in range(3):
            rep0 = ctx.tick_stats()["replayed"]
            el, n_warm = timed(lambda: ctx.svmpc_tick(state, w["n_iters"], want_outputs=False), ctx.sync)
            replayed_in_region = ctx.tick_stats()["replayed"] - rep0
            if replayed_in_region == 0:  # (a replay is executed by sync(), outside the clock: such a run is not a measurement)
                break
        extra["replayed_in_timed_region"] = replayed_in_region
        if replayed_in_region:
            extra["invalid"] = "ticks of the timed region were replayed outside it in all 3 attempts (device shared?): value is not a measurement"
        extra["warmup_ticks_run"] = n_warm
        extra["tick_paths"] = ctx.tick_stats()  # which kernel served the ticks (tick2 = owner-computes one-launch tick)
        # closed loop (simulations.py:104-123): optimize + forward -> chosen sequence -> its first action steps the plant -> next tick
        def closed_loop(serve):
            st = state.copy()
            if serve:
                ctx.serve_start(w["n_iters"], 2000.0)
            n_wcl, t0 = 0, time.perf_counter()
            while n_wcl < args.warmup or time.perf_counter() - t0 < MIN_WARM_S:  # the same warm-up rule, by time
                a_seq, _ = ctx.svmpc_tick(st, w["n_iters"], want_outputs="action")
                n_wcl += 1
            n_cl = max(args.steps, 1000)  # (a loop of 20 ticks is 2 ms; 200 ticks - 19 ms - read 9 770 once where five other runs read 10.4-10.65 k: 1 000)
            s0 = ctx.tick_stats()
            t0 = time.perf_counter()
            for _ in range(n_cl):
                a_seq, _ = ctx.svmpc_tick(st, w["n_iters"], want_outputs="action")
                st = pendulum_plant(st, a_seq[0, 0])
            el_cl = time.perf_counter() - t0
            s1 = ctx.tick_stats()
            if serve:
                ctx.serve_stop()
            return dict(ticks_per_s=n_cl / el_cl, us_per_tick=1e6 * el_cl / n_cl, ticks=n_cl, warmup_ticks_run=n_wcl,
                        served=s1["served"] - s0["served"], replayed=s1["replayed"] - s0["replayed"])

        if not args.open_loop_only:
            cl_plain = closed_loop(False)
            cl_served = closed_loop(True)
            extra["closed_loop_ticks_per_s"] = cl_served["ticks_per_s"]
            extra["closed_loop"] = dict(
                served=cl_served, unserved=cl_plain,
                note="every tick returns its chosen sequence to the host, whose first action steps a host pendulum plant; the new state feeds the "
                     "next tick.  served: dust_svmpc_serve_start - the outputs arrive through pinned host memory (no device-to-host copy, no stream "
                     "synchronisation) and the next tick is launched ahead of its plant state (bit-identical results: tests/test_gpu_serve.py); "
                     "unserved: one pinned device-to-host copy + one stream synchronisation per tick")
        ctx.close()
        workload = ("Pendulum N=%d, S=128, M=1, H=30, 5 SVGD iters, K1 (gpytorch-RBF) kernel, SGD, device Philox noise inside the tick; "
                    "one persistent kernel launch per tick" % w["N"])
        par = "single GPU"
        value = args.steps / el
        unit = "control steps/s"
    else:
        c4 = CFG4
        mu, theta = synth(c4["N"], c4["H"], 2, spread=1.0)
        common = dict(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
                      uncertain_params=("mass",), grid=particle_grid(), device=local, seed=1234)
        rngp = np.random.default_rng(5)
        params = (1.0 + 0.1 * rngp.standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
        st4 = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
        sh = ShardedSVMPC(common, rank, n_gpus, dist)
        sh.set_state(theta, mu)
        el, n_warm = timed(lambda: sh.tick(st4, c4["n_iters"], params=params), sh.sync)
        extra["warmup_ticks_run"] = n_warm
        comm_us = sh.ctx.comm_probe(c4["n_iters"], 50) if sh.c_side else None  # the tick's all-gathers alone (collective: every rank)
        extra["scale_workload"] = dict(workload="Particle N=16384, S=64, M=4, H=40, 1 SVGD iter, K1, SGD, device Philox noise (BASELINE configs[3])",
                                       n_gpus=n_gpus, ticks_per_s=args.steps / el, ms_per_tick=1e3 * el / args.steps, comm_us_per_tick=comm_us,
                                       ticks=args.steps, warmup_ticks_run=n_warm, pairwise_units_left_out=far_shares(sh.ctx))
        workload = ("Particle N=%d total (%d per GPU), S=64, M=4, H=40, 1 SVGD iter, K1 kernel, SGD, device Philox noise inside the tick"
                    % (c4["N"], c4["N"] // n_gpus))
        par = "particles sharded x%d (strong scaling), in-place RCCL all-gathers of score and theta per SVGD iteration" % n_gpus
        value = args.steps / el
        unit = "control steps/s (joint N=16384 problem)"
        # The same ticks with the three exchanges as DIRECT PEER STORES (dust_amd/csrc/peer_gather.hpp: every rank writes its piece into
        # every peer's IPC-mapped buffer, one arrival word per peer).  Tried after the library run, on the same contexts; accepted only
        # when the ranks still hold bit-identical particles afterwards (every rank keeps all N rows: a lost or stale piece shows up as a
        # disagreement).  `value` is the better of the two accepted figures; both are reported.
        if sh.c_side:
            # Every rank takes every branch below together: the set-up's verdict is collective (dust_comm_peer_gather), the two
            # validation ticks have no collective inside (peer stores with bounded waits: a rank whose pieces do not arrive raises
            # within seconds, it cannot hang the others), and the ranks AGREE (all_gather_object) before anything is timed.
            import zlib

            peer = dict(tried=True)
            err = None
            try:
                sh.ctx.comm_peer_gather(True)
            except Exception as e:  # noqa: BLE001 - collective outcome: every rank is here, or none
                err = "set-up: " + str(e)[:300]
            crc = None
            if err is None:
                try:
                    for _ in range(2):
                        sh.tick(st4, c4["n_iters"], params=params)
                    sh.sync()
                    th = sh.ctx.get_theta()
                    crc = (zlib.crc32(th.tobytes()), bool(np.isfinite(th).all()))
                except Exception as e:  # noqa: BLE001
                    err = "validation ticks: " + str(e)[:300]
            verdicts = [None] * n_gpus
            dist.all_gather_object(verdicts, (err, crc))
            errs = [v[0] for v in verdicts if v[0]]
            agree = not errs and all(v[1] == verdicts[0][1] for v in verdicts) and bool(verdicts[0][1][1])
            peer["ranks_agree"] = bool(agree)
            if errs:
                peer["error"] = errs[0]
            if agree:
                el_p, _ = timed(lambda: sh.tick(st4, c4["n_iters"], params=params), sh.sync)
                comm_p = sh.ctx.comm_probe(c4["n_iters"], 50)
                peer.update(ticks_per_s=args.steps / el_p, ms_per_tick=1e3 * el_p / args.steps, comm_us_per_tick=comm_p)
                if args.steps / el_p > value:
                    value = args.steps / el_p
                    el = el_p
                    par = "particles sharded x%d (strong scaling), direct peer-store all-gathers of score and theta per SVGD iteration" % n_gpus
            elif err is None or not err.startswith("set-up"):
                try:
                    sh.ctx.comm_peer_gather(False)  # back to the collective library for the legs below
                except Exception:  # noqa: BLE001
                    pass
            extra["scale_workload"]["peer_gather"] = peer
        # the SAME joint problem on ONE GPU (unsharded context, rank 0, outside the timed region; the other ranks wait): the figure a
        # strong-scaling efficiency of this line has to be computed against - `python bench.py --gpus 1` runs cfg2, not this
        if rank == 0 and (n_gpus > 1 or os.environ.get("DUST_BENCH_FORCE_DIST")):
            one = Context(**dict(common))
            one.set_theta(theta); one.set_prior(mu); one.set_a_mat(theta)
            # (the same number of warm-up and timed ticks as the sharded run: the particle set - and with it the share of exactly-zero
            #  kernel blocks the fused pairwise pass can skip - evolves with the ticks, so the two rates must be taken at the same age)
            for _ in range(n_warm):
                one.svmpc_tick(st4, c4["n_iters"], params=params, want_outputs=False)
            one.sync()
            t1 = time.perf_counter()
            n1 = args.steps
            for _ in range(n1):
                one.svmpc_tick(st4, c4["n_iters"], params=params, want_outputs=False)
            one.sync()
            extra["one_gpu_same_workload_ticks_per_s"] = n1 / (time.perf_counter() - t1)
            extra["scale_workload"]["one_gpu_ticks_per_s"] = extra["one_gpu_same_workload_ticks_per_s"]
            one.close()
        dist.barrier()
        # WEAK scaling in the particle count (north_star: "particle scaling"): 2048 particles PER GPU of the same Particle workload - at 8
        # GPUs this is configs[3] itself - against 2048 particles on one GPU (rank 0, unsharded, outside the timed region).  The pairwise
        # work per rank grows with the joint set (n_local x N pairs), so this is not free scaling either.
        try:
            n_w = 2048 * n_gpus
            if n_w != c4["N"] or n_gpus == 1:
                muw, thw = synth(n_w, c4["H"], 2, spread=1.0)
                shw = ShardedSVMPC(dict(common, N=n_w), rank, n_gpus, dist)
                shw.set_state(thw, muw)
                elw, _ = timed(lambda: shw.tick(st4, c4["n_iters"], params=params), shw.sync)
                tps_w = args.steps / elw
                shw.ctx.close()
            else:
                tps_w = args.steps / el
            one_w = None
            if rank == 0:
                mu1, th1 = synth(2048, c4["H"], 2, spread=1.0)
                o1 = Context(**dict(common, N=2048))
                o1.set_theta(th1); o1.set_prior(mu1); o1.set_a_mat(th1)
                for _ in range(40):
                    o1.svmpc_tick(st4, c4["n_iters"], params=params, want_outputs=False)
                o1.sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    o1.svmpc_tick(st4, c4["n_iters"], params=params, want_outputs=False)
                o1.sync()
                one_w = args.steps / (time.perf_counter() - t1)
                o1.close()
            dist.barrier()
            extra["weak_workload"] = dict(workload="Particle, 2048 particles per GPU (N = %d), S=64, M=4, H=40, 1 SVGD iter, K1" % n_w, n_particles_total=n_w,
                                          ticks_per_s=tps_w, one_gpu_2048_particles_ticks_per_s=one_w,
                                          particle_throughput_vs_one_gpu=(tps_w * n_w / (one_w * 2048.0)) if one_w else None)
        except Exception as e:  # noqa: BLE001
            extra["weak_workload"] = dict(error=str(e)[:300])
        if args.weak:
            n_tot = w["N"] * n_gpus
            mu2, th2 = synth(n_tot, w["H"], 1)
            common2 = dict(model=w["model"], N=n_tot, S=w["S"], M=1, H=w["H"], kernel=w["kernel"], lr=w["lr"], alpha=w["alpha"],
                           sigma_a=w["sigma_a"], sigma_p=w["sigma_p"], device=local, seed=1234)
            sh2 = ShardedSVMPC(common2, rank, n_gpus, dist)
            sh2.set_state(th2, mu2)
            el2, _ = timed(lambda: sh2.tick(state, w["n_iters"]), sh2.sync)
            extra["weak_cfg2"] = dict(joint_ticks_per_s=args.steps / el2, n_particles_total=n_tot,
                                      shard_ticks_per_s=args.steps / el2 * n_gpus)

    roofline = None
    if rank == 0 and not args.no_roofline and dist is None:
        rk = rk_early
        # The TIMED kernel on top: one launch of svmpc_tick2_kernel = one control tick = n_iters SVGD iterations + forward.
        # Algorithmic bytes per SURVEY 8d: B_roll per iteration (4 [S N D + N D + S N] + grad_lik out; the noise term counts although the
        # product draws it in registers: it is the traffic the contract's figure is defined over) + the epilogue 4 (S N + 4 N D).
        t_tick = el / args.steps
        b_iter = 4.0 * (w["S"] * w["N"] * w["H"] + 2 * w["N"] * w["H"] + w["S"] * w["N"])
        b_tick = w["n_iters"] * b_iter + 4.0 * (w["S"] * w["N"] + 4 * w["N"] * w["H"])
        ach = b_tick / t_tick / 1e9
        roofline = dict(kernel="dust::svmpc_tick2_kernel<0,1> (the timed kernel: one launch = one control tick)", bound="valu-issue/latency",
                        achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", hbm_frac=ach / HBM_PEAK_GBS, algorithmic_bytes_per_launch=b_tick,
                        avg_launch_us=1e6 * t_tick, launches=args.steps,
                        timing="the timed region itself: K back-to-back launches on the context's stream, wall clock between two device synchronisations")
        pj, psrc = profile_json("tick_pmc", "tick2")
        if pj and pj.get("SQ_INSTS_VALU_per_tick"):  # VALU issue fraction of the persistent tick kernel from the committed SQ counter pass
            insts = pj["SQ_INSTS_VALU_per_tick"]
            issue_s = insts / VALU_PEAK_WAVE_INSTR_PER_S
            roofline.update(frac=issue_s / t_tick, valu_wave_instructions_per_tick=insts, valu_issue_floor_us=issue_s * 1e6,
                            frac_definition="VALU issue floor (every vector instruction at the 2-cycle peak of its SIMD) / measured tick; measured "
                                            "issue prices on this chip (tools/valu_rate_probe.hip): plain VOP2 2.6, VOP3 / packed / DPP 4.3, "
                                            "transcendentals 8.2 cycles - priced per class the tick is ~0.75 issue-bound",
                            issue_source=psrc + " (rocprofv3 --pmc SQ_INSTS_VALU; source stamp %s)" % pj.get("source_stamp"))
        else:
            roofline.update(frac=ach / HBM_PEAK_GBS, frac_definition="hbm_frac (no current SQ_INSTS_VALU summary: %s)" % psrc)
        tj, tsrc = profile_json("tick_traffic", "tick2")
        roofline["traffic"] = tj["hbm_bytes_per_launch"] if tj else None
        roofline["traffic_source"] = (tsrc + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; source stamp %s)" % tj["source_stamp"]) if tj else tsrc
        forms = rk.pop("forms", None)
        roofline["rollout_kernel"] = rk   # the HBM-bound form of the rollout kernel (north_star: ">= 60 % of HBM roofline on the rollout kernel")
        roofline["forms"] = forms

    if rank == 0:
        # short keys first (a reader that keeps the head of the line keeps the figures); the long notes follow
        rkk = (roofline or {}).get("rollout_kernel") or {}
        out = {
            "metric": "MPC control steps/sec (SVGD-MPC tick: SVGD iterations + forward)",
            "value": value,
            "unit": unit,
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            # N > 1 shards ONE joint problem (BASELINE configs[3], north_star: "particle batches shard ... >= 6x particle scaling at 8 GPUs"):
            # strong scaling; the N = 1 line times configs[1] (the configuration the metric is quoted on) and carries the N = 1 point of
            # the sharded workload in `scale_workload`
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "closed_loop_ticks_per_s": extra.get("closed_loop_ticks_per_s"),
            "rollout_kernel_hbm_frac": rkk.get("frac"),
            "rollout_kernel_achieved_gbs": rkk.get("achieved"),
            "rollout_kernel_avg_launch_us": rkk.get("avg_launch_us"),
            "rollout_kernel_traffic_bytes": rkk.get("traffic"),
            "cfg4_one_gpu_ms_per_tick": (extra.get("scale_workload") or {}).get("ms_per_tick"),
            "config": {"workload": workload, "parallelism": par},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        out.update(extra)
        out["config"]["other_configs"] = ("BASELINE configs[4] (cfg5) is run with M = 8 dynamics samples per rollout (the yaml's params_samples), drawn from "
                                          "the 256-particle MPF (tools/configs_bench.py -> profiles/round6_configs.json).  SURVEY 8d's alternative, M = 256 "
                                          "(every dynamics particle once), is accepted by the launch-per-iteration rollout kernels (any M; parity 9e-8 vs the "
                                          "oracle) but not by the one-launch tick (T2_MAXM = 64): 114 ticks/s (tools/cfg5_m256.py)")
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
