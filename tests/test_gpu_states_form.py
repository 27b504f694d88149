"""The whole-line stored-states kernel (dust_amd/csrc/rollout_states.hpp) against the oracle and against the per-particle
staging kernel it replaces (DUST_STATES_FORM=0): same costs, same states, bit for bit between the two device kernels.

MultiDISCO.forward returns `states` [M][S][N][H+1][ds] (dust/controllers/disco.py:394); the Particle family at fp32 takes the
whole-line form when 8-particle groups are whole 128-byte lines (N % 8 == 0, H + 1 odd and >= 9) and M is even; the Pendulum
family when 16-particle groups are (N % 16 == 0, H + 1 odd, H >= 16)."""
import os

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _run(N, S, M, H, can_crash, with_obstacle, poison=None, state=None, seed=0):
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    rng = np.random.default_rng(seed + 7 * N + S)
    grid = grid_4x4_map() if with_obstacle else None
    up = ("mass",) if M > 1 else None
    kw = dict(model="particle", N=N, S=S, M=M, H=H, uncertain_params=up, can_crash=can_crash, with_obstacle=with_obstacle)
    actions = (1.5 * rng.standard_normal((S, N, H, 2))).astype(np.float32)
    if poison is not None:
        actions[poison] = np.nan
    params = None if up is None else rng.uniform(0.6, 1.6, (M, 1)).astype(np.float32)
    st = np.array([-5.2, -7.3, 4.0, 3.0] if state is None else state, np.float32)
    o = Oracle(grid=grid, **kw)
    ref_costs, ref_states = o.rollout_cost(st, actions, params, want_states=True)
    out = {}
    for form in ("1", "0"):
        os.environ["DUST_STATES_FORM"] = form
        try:
            c = Context(grid=grid, kernel="K1", alpha=1e-4, sigma_a=5.0, sigma_p=5.0, **kw)
            c.set_a_mat(np.zeros((N, H, 2), np.float32))
            c.profile(True)
            costs, states, _, omega = c.disco_forward(st, actions, params, want_states=True)
            used = "states_kernel" in c.profile_get()
            amix = c.get_a_mix()
            c.close()
        finally:
            os.environ.pop("DUST_STATES_FORM", None)
        out[form] = (costs, states, omega, used)
        out[form + "_amix"] = amix
    return ref_costs, ref_states, out


@pytest.mark.parametrize("N,S,M,H,can_crash,with_obstacle", [
    (16, 12, 4, 10, True, True),     # ragged S (12 = 8 + 4), two waves
    (8, 8, 2, 8, False, False),      # smallest legal shape, no map
    (24, 64, 8, 40, False, True),    # cfg3's S / H, four waves, obstacles that do not stop the particle
    (32, 17, 16, 12, True, True),    # two pair iterations per wave
])
def test_whole_line_states_vs_oracle_and_staged_kernel(N, S, M, H, can_crash, with_obstacle):
    ref_costs, ref_states, out = _run(N, S, M, H, can_crash, with_obstacle)
    costs, states, omega, used = out["1"]
    costs0, states0, omega0, used0 = out["0"]
    assert used and not used0  # the new kernel really ran (and the switch really selects the old one)
    assert relerr(costs, ref_costs) < TOL
    assert relerr(states, ref_states) < TOL
    assert np.array_equal(states, states0)  # every byte of every line, heads and tails included
    assert np.array_equal(costs, costs0)
    assert np.array_equal(omega, omega0)
    # the two-pass form hands its costs to the regular kernel: the likelihood record (a_mix = softmax(eta), disco.py:393) must be
    # refreshed as after a one-pass sample (round 3: it was left stale)
    assert np.allclose(out["1_amix"], out["0_amix"], rtol=1e-4, atol=1e-7) and abs(float(out["1_amix"].sum()) - 1.0) < 1e-4


def test_whole_line_states_general_path_on_nan_action():
    """A NaN action sends that workgroup down the reference-order step functions (torch.clamp propagates NaN, v_med3 would not):
    same bits as the staged kernel's general loop, NaNs in the same places."""
    ref_costs, ref_states, out = _run(16, 16, 4, 10, True, True, poison=(3, 5, 2, 1))
    costs, states, omega, used = out["1"]
    costs0, states0, omega0, _ = out["0"]
    assert used
    assert np.array_equal(np.isnan(states), np.isnan(ref_states))
    assert np.array_equal(states, states0, equal_nan=True)
    assert np.array_equal(costs, costs0, equal_nan=True)
    ok = ~np.isnan(ref_costs)
    assert relerr(costs[ok], ref_costs[ok]) < TOL


def test_whole_line_states_not_taken_when_lines_do_not_close():
    """H + 1 even (rows of 16 (H+1) bytes pair up into lines differently) or N % 8 != 0: the staged kernel runs, results as before."""
    for N, H in ((16, 11), (12, 10)):
        ref_costs, ref_states, out = _run(N, 8, 2, H, True, True)
        costs, states, _, used = out["1"]
        assert not used
        assert relerr(states, ref_states) < TOL and relerr(costs, ref_costs) < TOL


def _run_pend(N, S, M, H, poison=None, seed=0):
    from dust_amd import Context
    from oracle import Oracle

    rng = np.random.default_rng(seed + 5 * N + S)
    up = ("length", "mass") if M > 1 else None
    kw = dict(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=up)
    actions = (1.5 * rng.standard_normal((S, N, H, 1))).astype(np.float32)
    if poison is not None:
        actions[poison] = np.nan
    params = None if up is None else rng.uniform(0.6, 1.4, (M, 2)).astype(np.float32)
    st = np.array([3.0, -0.4], np.float32)
    o = Oracle(**kw)
    ref_costs, ref_states = o.rollout_cost(st, actions, params, want_states=True)
    out = {}
    for form in ("1", "0"):
        os.environ["DUST_STATES_FORM"] = form
        try:
            c = Context(kernel="K1", sigma_a=2.0, sigma_p=2.0, **kw)
            c.set_a_mat(np.zeros((N, H, 1), np.float32))
            c.profile(True)
            costs, states, _, omega = c.disco_forward(st, actions, params, want_states=True)
            used = "states_kernel" in c.profile_get()
            c.set_a_mat(np.zeros((N, H, 1), np.float32))
            lean_costs = c.disco_forward(st, actions, params, want_states=False)[0]  # the rollout kernel without stored states
            c.close()
        finally:
            os.environ.pop("DUST_STATES_FORM", None)
        out[form] = (costs, states, omega, used, lean_costs)
    return ref_costs, ref_states, out


@pytest.mark.parametrize("N,S,M,H", [
    (16, 16, 1, 30),   # cfg2's horizon: 248-byte rows, 31 lines per 16-particle group
    (32, 21, 3, 16),   # shortest legal horizon, ragged S, three dynamics samples in sequence
    (48, 128, 8, 30),  # cfg5's S / M / H
    (16, 5, 2, 38),
])
def test_pendulum_whole_line_states_vs_oracle_and_staged_kernel(N, S, M, H):
    """The Pendulum form of the whole-line kernel (16-particle groups, 16 eight-byte slots per line, heads held in registers).  It
    runs the branch-free trig path of the rollout kernel WITHOUT stored states (the staged kernel runs the reference-order step
    functions): costs bit-equal to that kernel's, states within an ulp or two of the staged kernel's."""
    ref_costs, ref_states, out = _run_pend(N, S, M, H)
    costs, states, omega, used, lean = out["1"]
    costs0, states0, omega0, used0, _ = out["0"]
    assert used and not used0
    assert relerr(costs, ref_costs) < TOL
    assert relerr(states, ref_states) < TOL
    assert np.array_equal(costs, lean)
    assert relerr(states, states0) < 2e-6 and relerr(costs, costs0) < 2e-6 and relerr(omega, omega0) < 1e-4


def test_pendulum_whole_line_states_general_path_and_fallbacks():
    ref_costs, ref_states, out = _run_pend(16, 16, 2, 20, poison=(2, 7, 3, 0))
    costs, states, _, used, _ = out["1"]
    costs0, states0, _, _, _ = out["0"]
    assert used
    assert np.array_equal(np.isnan(states), np.isnan(ref_states))
    assert np.array_equal(states, states0, equal_nan=True) and np.array_equal(costs, costs0, equal_nan=True)
    for N, H in ((16, 15), (16, 31), (24, 30)):  # too short / H + 1 even / N % 16 != 0: the staged kernel
        ref_costs, ref_states, out = _run_pend(N, 8, 1, H)
        costs, states, _, used, _ = out["1"]
        assert not used
        assert relerr(states, ref_states) < TOL and relerr(costs, ref_costs) < TOL


def _run_pend_f16(N, S, M, H, poison=None, seed=0):
    from dust_amd import Context
    from oracle import Oracle

    rng = np.random.default_rng(seed + 7 * N + S)
    up = ("length", "mass") if M > 1 else None
    kw = dict(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=up)
    actions = (1.5 * rng.standard_normal((S, N, H, 1))).astype(np.float32)
    if poison is not None:
        actions[poison] = np.nan
    params = None if up is None else rng.uniform(0.6, 1.4, (M, 2)).astype(np.float32)
    st = np.array([3.0, -0.4], np.float32)
    ref_costs, ref_states = Oracle(**kw).rollout_cost(st, actions, params, want_states=True)
    out = {}
    for form in ("1", "0"):
        os.environ["DUST_STATES_FORM"] = form
        try:
            c = Context(kernel="K1", sigma_a=2.0, sigma_p=2.0, **kw)
            c.set_a_mat(np.zeros((N, H, 1), np.float32))
            c.profile(True)
            costs, states, _, omega = c.disco_forward(st, actions, params, want_states=True, store_f16=True)
            used = "states_kernel" in c.profile_get()
            c.set_a_mat(np.zeros((N, H, 1), np.float32))
            lean_costs = c.disco_forward(st, actions, params, want_states=False)[0]
            c.close()
        finally:
            os.environ.pop("DUST_STATES_FORM", None)
        out[form] = (costs, states, omega, used, lean_costs)
    return ref_costs, ref_states, out


@pytest.mark.parametrize("N,S,M,H", [
    (32, 8, 1, 30),    # cfg2's horizon: 124-byte rows, 31 lines per 32-particle group
    (64, 21, 3, 16),   # ragged S, three dynamics samples in sequence
    (64, 128, 8, 30),  # cfg5's S / M / H ("fp16 rollout")
    (32, 5, 2, 37),    # H + 1 even: two-way LDS conflicts in the image, same bytes
    (96, 9, 1, 5),
])
def test_pendulum_binary16_whole_line_states_vs_oracle_and_staged_kernel(N, S, M, H):
    """DUST_STORE_F16 (cfg5's "fp16 rollout": storage only, fp32 arithmetic) through the whole-line form: a trajectory is 4 (H+1)
    bytes, shorter than a line, so a wave images 2 samples x 32 adjacent particles in LDS and copies H+1 whole lines per sample out.
    States: the oracle's fp32 states rounded to binary16 (one ulp of slack, plus the fp32 difference underneath), and the staged
    kernel's halves up to the same; costs bit-equal to the rollout kernel without stored states."""
    ref_costs, ref_states, out = _run_pend_f16(N, S, M, H)
    costs, states, omega, used, lean = out["1"]
    costs0, states0, omega0, used0, _ = out["0"]
    assert used and not used0
    assert states.dtype == np.float16 and states.shape == ref_states.shape
    assert relerr(costs, ref_costs) < TOL and np.array_equal(costs, lean)
    ref16 = ref_states.astype(np.float16).astype(np.float32)
    s32, s032 = states.astype(np.float32), states0.astype(np.float32)
    # one binary16 ulp, plus the fp32 difference underneath (branch-free trig path vs reference-order steps: 2e-6 of the largest
    # state, the bound of the fp32 test above - it shows where a velocity passes through zero)
    ulp16 = np.maximum(np.abs(ref16), 2.0 ** -14) * 2.0 ** -10 + 2e-6 * np.abs(ref_states).max()
    assert np.all(np.abs(s32 - ref16) <= ulp16) and np.mean(s32 != ref16) < 1e-3
    assert np.all(np.abs(s32 - s032) <= ulp16) and np.mean(s32 != s032) < 1e-3
    # omega = softmax_S(-costs / temp): one fp32 ulp of a cost of 6e3 is 5e-4 in a logit, so the weights of two kernels whose costs
    # differ in the last place (branch-free trig path vs reference-order steps) can only be held to a few of those
    ulp_logit = float(np.spacing(np.float32(np.abs(costs0).max())))
    assert relerr(costs, costs0) < 2e-6 and relerr(omega, omega0) < max(1e-4, 2.0 * ulp_logit)


def test_pendulum_binary16_whole_line_states_general_path_and_fallbacks():
    ref_costs, ref_states, out = _run_pend_f16(32, 8, 2, 20, poison=(2, 7, 3, 0))  # one workgroup: all of it takes the general instance
    costs, states, _, used, _ = out["1"]
    costs0, states0, _, _, _ = out["0"]
    assert used
    assert np.array_equal(np.isnan(states), np.isnan(ref_states))
    assert np.array_equal(states, states0, equal_nan=True) and np.array_equal(costs, costs0, equal_nan=True)
    ref_costs, ref_states, out = _run_pend_f16(48, 8, 1, 30)  # N % 32 != 0: the staged kernel
    costs, states, _, used, _ = out["1"]
    assert not used
    assert relerr(states.astype(np.float32), ref_states) < 2e-3 and relerr(costs, ref_costs) < TOL


def _run_particle_f16(N, S, M, H, can_crash, with_obstacle, poison=None, seed=0):
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    rng = np.random.default_rng(seed + 11 * N + S)
    grid = grid_4x4_map() if with_obstacle else None
    up = ("mass",) if M > 1 else None
    kw = dict(model="particle", N=N, S=S, M=M, H=H, uncertain_params=up, can_crash=can_crash, with_obstacle=with_obstacle)
    actions = (1.5 * rng.standard_normal((S, N, H, 2))).astype(np.float32)
    if poison is not None:
        actions[poison] = np.nan
    params = None if up is None else rng.uniform(0.6, 1.6, (M, 1)).astype(np.float32)
    st = np.array([-5.2, -7.3, 4.0, 3.0], np.float32)
    ref_costs, ref_states = Oracle(grid=grid, **kw).rollout_cost(st, actions, params, want_states=True)
    out = {}
    for form in ("1", "0"):
        os.environ["DUST_STATES_FORM"] = form
        try:
            c = Context(grid=grid, kernel="K1", alpha=1e-4, sigma_a=5.0, sigma_p=5.0, **kw)
            c.set_a_mat(np.zeros((N, H, 2), np.float32))
            c.profile(True)
            costs, states, _, omega = c.disco_forward(st, actions, params, want_states=True, store_f16=True)
            used = "states_kernel" in c.profile_get()
            c.close()
        finally:
            os.environ.pop("DUST_STATES_FORM", None)
        out[form] = (costs, states, omega, used)
    return ref_costs, ref_states, out


@pytest.mark.parametrize("N,S,M,H,can_crash,with_obstacle", [
    (16, 4, 2, 16, False, False),    # smallest legal shape: one group, one pair, 17 rows (two lines complete in the loop)
    (32, 9, 4, 40, True, True),      # cfg3's H, ragged S (9 = 4 + 4 + 1), two waves, crash semantics
    (48, 64, 8, 40, False, True),    # cfg3's S / H, four waves, three groups
    (16, 6, 16, 18, True, True),     # two pair iterations per wave
])
def test_particle_binary16_whole_line_states_vs_oracle_and_staged_kernel(N, S, M, H, can_crash, with_obstacle):
    """DUST_STORE_F16 for the Particle family through a whole-line kernel of its own (round 4): 8-byte states, groups of 16 adjacent
    particles = H + 1 whole lines.  States: the oracle's fp32 states rounded to binary16 and the staged kernel's halves BIT FOR BIT
    (same step functions, same conversions); costs bit-equal to the staged kernel's."""
    ref_costs, ref_states, out = _run_particle_f16(N, S, M, H, can_crash, with_obstacle)
    costs, states, omega, used = out["1"]
    costs0, states0, omega0, used0 = out["0"]
    assert used and not used0
    assert states.dtype == np.float16 and states.shape == ref_states.shape
    assert relerr(costs, ref_costs) < TOL
    assert np.array_equal(states, states0)  # every byte of every line, heads and tails included
    assert np.array_equal(costs, costs0) and np.array_equal(omega, omega0)
    ref16 = ref_states.astype(np.float16).astype(np.float32)
    s32 = states.astype(np.float32)
    ulp16 = np.maximum(np.abs(ref16), 2.0 ** -14) * 2.0 ** -10 + 2e-6 * np.abs(ref_states).max()
    assert np.all(np.abs(s32 - ref16) <= ulp16) and np.mean(s32 != ref16) < 1e-3


def test_particle_binary16_whole_line_states_general_path_and_fallbacks():
    ref_costs, ref_states, out = _run_particle_f16(16, 8, 4, 20, True, True, poison=(3, 5, 2, 1))  # a NaN action: the general instance
    costs, states, _, used = out["1"]
    costs0, states0, _, _ = out["0"]
    assert used
    assert np.array_equal(np.isnan(states), np.isnan(ref_states))
    assert np.array_equal(states, states0, equal_nan=True) and np.array_equal(costs, costs0, equal_nan=True)
    for N, H in ((24, 20), (16, 21), (16, 10)):  # N % 16 != 0, H + 1 even, H < 16: the staged kernel
        ref_costs, ref_states, out = _run_particle_f16(N, 8, 2, H, True, True)
        costs, states, _, used = out["1"]
        assert not used
        assert relerr(states.astype(np.float32), ref_states) < 2e-3 and relerr(costs, ref_costs) < TOL


@pytest.mark.parametrize("f16", [False, True])
def test_full_size_stored_states_sampled_vs_oracle(f16):
    """The stored-states rollout kernel AT THE SHAPE bench.py TIMES IT ON (BASELINE configs[2]: Particle N = 4096, S = 64, M = 64, H = 40:
    16.8 M trajectories, 2.77 G state values = 11.0 GB in fp32 / 5.5 GB in binary16 - beyond 32-bit element indexing, beyond 2^33 bytes;
    VERDICT r4: the kernel was only ever checked at N <= 32, where no address needs more than 32 bits).  The states stay on the device
    (dust_likelihood_sample with DUST_STORE_STATES); 96 sampled trajectories (m, s, n) - a third of them beyond byte offset 2^33 in
    fp32 (2^32 in binary16), the last row, the rows either side of element index 2^31 / 2^32 and of byte offset 2^32 / 2^33 - are
    fetched with dust_get_states_rows and compared (a) with the oracle's states of the same rollouts at 1e-5 (binary16: the oracle's
    value rounded to binary16, 1 ulp) and (b) BIT FOR BIT with the staging kernel (DUST_STATES_FORM=0) on the same rows.
    Reference: disco.py:190-200 (states), 394 (returned)."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    N, S, M, H = 4096, 64, 64, 40
    rng = np.random.default_rng(4096)
    mu = rng.standard_normal((N, H, 2)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, 2))).astype(np.float32)
    eps = rng.standard_normal((S, N, H, 2)).astype(np.float32)
    state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    params = (1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
    grid = grid_4x4_map()
    R, row_elems, eb = M * S * N, (H + 1) * 4, (2 if f16 else 4)
    row_bytes = row_elems * eb
    # sampled rollouts: the oracle runs a 48-particle subset with all (m, s); rows are drawn among its trajectories
    idx = np.sort(rng.choice(N, 48, replace=False))
    idx[0], idx[-1] = 0, N - 1
    pick = []
    edge_rows = [0, R - 1]
    for boundary_elems in (2 ** 31, 2 ** 32):
        edge_rows += [boundary_elems // row_elems - 1, boundary_elems // row_elems, boundary_elems // row_elems + 1]
    for boundary_bytes in (2 ** 32, 2 ** 33):
        edge_rows += [boundary_bytes // row_bytes - 1, boundary_bytes // row_bytes, boundary_bytes // row_bytes + 1]
    edge_rows = sorted({r for r in edge_rows if 0 <= r < R})
    # (edge rows name arbitrary particles: they are checked against the staging kernel bit for bit, and against the oracle when their
    #  particle is in the subset - n is forced into it below by replacing subset members)
    edge_n = sorted({r % N for r in edge_rows})
    free = [i for i in range(1, 47) if idx[i] not in edge_n]
    for j, n in enumerate(x for x in edge_n if x not in idx):
        idx[free[j]] = n
    idx = np.sort(idx)
    assert len(set(idx)) == 48 and all(n in idx for n in edge_n)
    far = (2 ** 33 if not f16 else 2 ** 32) // row_bytes  # rows from here on start beyond the byte offset
    for i in range(96 - len(edge_rows)):
        m = int(rng.integers(far // (S * N) + 1, M)) if i % 3 == 0 else int(rng.integers(0, M))
        pick.append((m * S + int(rng.integers(0, S))) * N + int(idx[rng.integers(0, 48)]))
    rows = np.array(sorted(set(edge_rows + pick)), np.int64)
    assert (rows >= far).sum() * 3 >= len(rows) and rows.max() == R - 1
    o = Oracle(model="particle", N=48, S=S, M=M, H=H, uncertain_params=("mass",), grid=grid)
    sg = np.full(2, 5.0, np.float32)
    actions = o.sample_actions(theta[idx], np.ascontiguousarray(eps[:, idx]), sg)
    ref_costs, ref_states = o.rollout_cost(state, actions, params, want_states=True)  # [M][S][48][H+1][4]
    pos = {int(n): i for i, n in enumerate(idx)}
    want = np.stack([ref_states[r // (S * N), (r // N) % S, pos[int(r % N)]] for r in rows])
    got = {}
    for form in ("1", "0"):
        os.environ["DUST_STATES_FORM"] = form
        try:
            c = Context(model="particle", N=N, S=S, M=M, H=H, uncertain_params=("mass",), grid=grid, kernel="K1", lr=0.5, alpha=1e-4,
                        sigma_a=5.0, sigma_p=5.0)
            c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
            c.profile(True)
            costs = c.likelihood_sample(state, eps, params, store_states=True, store_f16=f16)
            used = any("states" in k for k in c.profile_get())
            got[form] = (c.get_states_rows(rows, f16=f16), costs[:, idx], used)
            c.close()
        finally:
            os.environ.pop("DUST_STATES_FORM", None)
    assert got["1"][2] and not got["0"][2], "the whole-line kernel served form 1, the staging kernel form 0"
    st1, st0 = got["1"][0], got["0"][0]
    assert st1.shape == (len(rows), H + 1, 4)
    assert np.array_equal(st1.view(np.uint16 if f16 else np.uint32), st0.view(np.uint16 if f16 else np.uint32)), "bit for bit vs the staging kernel"
    if f16:
        # binary16 is the STORAGE format: the device rounds its fp32 state once; the oracle's fp32 state rounded the same way agrees to
        # 1 ulp of binary16 wherever the fp32 values agree to 1e-5
        w16 = want.astype(np.float16)
        ulp = np.abs(np.spacing(w16.astype(np.float32).astype(np.float16))).astype(np.float32)
        assert np.all(np.abs(st1.astype(np.float32) - w16.astype(np.float32)) <= ulp + 1e-12)
    else:
        assert relerr(st1, want) < TOL
        assert float(np.abs(st1.astype(np.float64) - want).max()) < 1e-4
    assert relerr(got["1"][1], ref_costs) < TOL and np.array_equal(got["1"][1], got["0"][1])
