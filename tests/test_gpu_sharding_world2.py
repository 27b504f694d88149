"""The C-side sharded tick (dust_comm_init; dust_amd.hip sharded_steps / sharded_forward) at WORLD SIZES 2 AND 4, as separate PROCESSES,
on the one GPU of the test box - through tests/fake_rccl, a stand-in collective library over HIP IPC (DUST_RCCL_LIB).

SURVEY 8e / BASELINE north_star: particles sharded over the ranks, an all-gather of the score rows before the pairwise (Stein) step and
of the particles after the update, one of the log-weights per tick.  The real RCCL needs a GPU per rank, so until round 5 this code had
only run at world 1 ("collective ordering across the main and side stream is exactly the kind of bug that appears only at
world 2").  Here every rank is the product's ShardedSVMPC(c_side=True) in its own process; the stand-in synchronises ONLY the stream
each collective is issued on, so a missing cross-stream dependency shows up as stale data.  Both stream orders: the particle
all-gather on the side stream under the next iteration's rollouts (default) and everything on one stream (DUST_NO_COMM_OVERLAP).
Result: the same best particle, a_seq and the final particles equal to the unsharded tick to 2e-6 element-wise (the sharded passes sum
key slices in another order), weights to 1e-5 - the bounds of the single-process LocalComm tests; the ranks agree bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import elemerr, relerr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_SRC = os.path.join(HERE, "fake_rccl", "fake_rccl.cpp")
FAKE_SO = os.path.join(HERE, "fake_rccl", "libfakerccl.so")


@pytest.fixture(scope="module")
def fake_rccl():
    if not os.path.exists(FAKE_SO) or os.path.getmtime(FAKE_SO) < os.path.getmtime(FAKE_SRC):
        subprocess.run(["g++", "-shared", "-fPIC", "-O1", FAKE_SRC, "-I/opt/rocm/include", "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", FAKE_SO],
                       check=True)
    return FAKE_SO


CASES = [
    dict(model="pendulum", N=256, S=64, M=1, H=12, K=3, T=3, ext_noise=True),
    dict(model="particle", N=128, S=64, M=4, H=40, K=2, T=3, ext_noise=True),                 # D = 80 (the cfg4 row shape), sampled dynamics
    dict(model="pendulum", N=512, S=32, M=1, H=10, K=2, T=2, ext_noise=False),                # device Philox noise, keyed by the global particle index
    dict(model="particle", N=2048, S=16, M=1, H=20, K=1, T=2, ext_noise=True),                # the large-set pairwise kernels (N >= 2048)
    dict(model="pendulum", N=128, S=64, M=1, H=10, K=2, T=2, ext_noise=True, optimizer="Adam"),
]


@pytest.mark.parametrize("gather", ["library", "peer", "peer-refused"])
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_c_side_sharded_tick_in_separate_processes(ci, world, overlap, gather, fake_rccl, tmp_path):
    """gather = "library": the three exchanges through the collective library's all-gather; "peer": as direct peer stores into
    IPC-mapped buffers with arrival words (dust_amd/csrc/peer_gather.hpp, DUST_PEER_GATHER=1) - the same results, bit for bit between
    the ranks, on both stream orders (with the overlap the particle pieces travel under the next iteration's rollouts and their
    arrival is awaited in front of the prior pass)."""
    case = CASES[ci]
    if world == 4 and ci in (1, 4):
        pytest.skip("world 4 runs on three of the cases")
    if gather == "peer-refused" and (ci, world, overlap) != (0, 2, True):
        pytest.skip("one case: a rank that cannot map its peers (test hook) - EVERY rank must fall back to the library's all-gathers, none may hang")
    sys.path.insert(0, HERE)
    from sharded_worker import case_inputs

    cj = tmp_path / "case.json"
    cj.write_text(json.dumps(case))
    env = dict(os.environ, DUST_RCCL_LIB=fake_rccl)
    # the ranks share ONE GPU here: kernels that spin on their own workgroups (the fused launch forms) would meet the other process's
    # grid (handoff.hpp); the sharded tick's collective logic is what this test is about
    env["DUST_NO_FUSE"] = "1"
    env.pop("DUST_NO_COMM_OVERLAP", None)
    if not overlap:
        env["DUST_NO_COMM_OVERLAP"] = "1"
    env.pop("DUST_PEER_GATHER", None)
    env.pop("DUST_PEER_TEST_FAIL", None)
    if gather in ("peer", "peer-refused"):
        env["DUST_PEER_GATHER"] = "1"
    if gather == "peer-refused":
        env["DUST_PEER_TEST_FAIL"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), str(r), str(world), str(tmp_path), str(cj)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=240)
            outs.append(o)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    res = [dict(np.load(tmp_path / ("out_%d.npz" % r))) for r in range(world)]
    # the unsharded tick, in this process (the workers have gone: the device is free again)
    from dust_amd import Context

    kw, mu, th, state, eps, params = case_inputs(case)
    saved = os.environ.get("DUST_NO_PERSIST")
    os.environ["DUST_NO_PERSIST"] = "1"  # launch-per-iteration path: the kernels the sharded tick runs
    try:
        ref = Context(**kw)
        ref.set_theta(th); ref.set_prior(mu); ref.set_a_mat(th)
        want = [ref.svmpc_tick(state, case["K"], None if eps is None else eps[t], None if params is None else params[t]) for t in range(case["T"])]
        rt, ra = ref.get_theta(), ref.get_a_mat()
        ref.close()
    finally:
        os.environ.pop("DUST_NO_PERSIST", None)
        if saved is not None:
            os.environ["DUST_NO_PERSIST"] = saved
    n_loc = case["N"] // world
    for r in range(world):
        for t in range(case["T"]):
            # (the chosen sequence is a particle row: same particle, and the row to the particles' own bound - with the overlap the score
            #  is formed in prior_finish_kernel instead of the rollout kernel's merge epilogue: another order of the slice sums, an ulp)
            assert elemerr(res[r]["a_seq"][t], want[t][0]) < 2e-6, (r, t, np.abs(res[r]["a_seq"][t] - want[t][0]).max())
            assert int(np.argmax(res[r]["pw"][t])) == int(np.argmax(want[t][1])), (r, t)
            assert relerr(res[r]["pw"][t], want[t][1]) < 1e-5, (r, t)
        assert elemerr(res[r]["theta"], rt) < 2e-6, r                         # every rank holds ALL particles after the tick's gathers
        rows = slice(r * n_loc, (r + 1) * n_loc)
        assert elemerr(res[r]["a_mat"][rows], ra[rows]) < 1e-5, r              # a_mat rows are rank-local state
    for r in range(1, world):
        assert np.array_equal(res[r]["theta"], res[0]["theta"]), "the ranks disagree about the particles"
