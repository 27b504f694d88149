"""GPU parity tests proper: the HIP path, called through the C ABI (dust_amd.backend.Context -> libdust_amd.so), against
(1) golden vectors produced by the reference itself and (2) the CPU oracle on seeded inputs.

Tolerance: BASELINE.json's north_star states 1e-5 relative fp32 on identical seeds; each assertion carries its number.
Stages downstream of the costs are fed the REFERENCE's costs/actions (SURVEY.md "tolerance amplification": costs are
O(1e3) and enter softmax(-alpha c), so one ulp of a cost is already 2e-4 relative on a weight)."""
import numpy as np
import pytest

import os

from helpers import feed_ctrl_noise, tick2_ticks_expected, elemerr, is_adam, relerr, scenario_kwargs
from test_oracle_golden import K1_F64_CASES, SVMPC_CASES, _prior_at, k1_tolerance

pytestmark = pytest.mark.gpu
TOL = 1e-5


def ctx_kwargs(g):
    kw = scenario_kwargs(g)
    sa, sp = np.asarray(g["sigma_a"], np.float32).reshape(-1), np.asarray(g["sigma_p"], np.float32).reshape(-1)  # (a_cov / p_cov in kw win)
    kw.update(kernel=str(g["kernel_kind"]), likelihood=str(g["lik_kind"]), lr=float(g["lr"]), alpha=float(g["alpha"]),
              temperature=float(g["temperature"]), sigma_a=float(sa[0]) if sa.size == 1 else sa, sigma_p=float(sp[0]) if sp.size == 1 else sp,
              weighted_prior=bool(int(g["weighted_prior"])), roll_strategy=str(g["roll_strategy"]))
    a_reg, temp = float(g["a_reg"]), float(g["temperature"])
    kw["ctrl_penalty"] = 1.0 - a_reg / temp
    if "deterministic" in g:  # Particle(deterministic=) particle.py:31 (round-5 fixtures)
        kw["deterministic"] = bool(int(g["deterministic"]))
    if "k2_bandwidth" in g and float(g["k2_bandwidth"]) >= 0:
        kw["k2_bandwidth"] = float(g["k2_bandwidth"])  # iid_mp(RBF(bandwidth >= 0)): fixed bandwidth
    if "k2_minimum_bw" in g:
        kw["k2_minimum_bw"] = float(g["k2_minimum_bw"])  # RBF(minimum_bw=): the clamp of the median-trick bandwidths
    return kw


def k2_tolerance(theta, h, shared, da):
    """K2's reference side (dust/kernels/base_kernels.py:53-89) forms (x_i - x_j)^2 as -2XY + XX + YY in fp32: cancellation noise
    ~ 4 eps x^2 on every pair distance, i.e. that much over h RELATIVE noise on the kernel values.  The oracle follows the
    reference's formula (and meets 1e-5); the HIP kernel uses exact differences and can only agree to that bound."""
    x = np.asarray(theta, np.float64).reshape(theta.shape[0], -1)
    x2 = (x * x).max(0)  # per flattened dimension
    if shared:
        x2 = x2.reshape(-1, da).sum(1)
    return max(TOL, 4 * 6e-8 * float((x2 / np.asarray(h, np.float64)).max()))


def make_ctx(g, name=""):
    from dust_amd import Context
    from oracle import grid_4x4_map

    kw = ctx_kwargs(g)
    if is_adam(name):
        kw["optimizer"] = "Adam"
    grid = grid_4x4_map() if kw["model"] == "particle" else None
    c = Context(grid=grid, **kw)
    return c


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_rollout_costs_vs_reference(golden, name):
    """a1-a6: policy noise, rollouts, costs, MPPI side effects - against the reference's own outputs."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    theta, a_mat = g["theta0"], g["a_mat0"]
    for t in range(T):
        for k in range(K):
            c.set_theta(theta)
            c.set_a_mat(a_mat)
            params = g["params"][t, k] if "params" in g else None
            feed_ctrl_noise(c, g, t, k)
            costs, actions = c.likelihood_sample(g["state"][t, k], g["eps"][t, k], params, want_actions=True)
            if "a_cov" in g:  # (a row of L eps is a two-term sum: agrees with torch's matmul to an ulp)
                assert elemerr(actions, g["actions"][t, k]) < 1e-6
            else:
                assert np.array_equal(actions, g["actions"][t, k]), "a1 must be bit-exact"
            assert elemerr(costs, g["costs"][t, k]) < TOL, (name, t, k)
            assert relerr(c.get_a_mat(), g["omega_amat"][t, k]) < 1e-4  # omega = softmax of O(1e3) logits (see module doc)
            # a_mix = softmax_n(logsumexp_s(-c/temp)): one fp32 ulp of a cost moves a logit by ulp(c)/temp, so the check is
            # meaningful only while that is small (Particle costs are O(1e7): ulp = 2-4, i.e. factors of e^2 in the
            # reference's own result)
            ulp_logit = float(np.spacing(np.float32(np.abs(g["costs"][t, k]).max()))) / float(g["temperature"])
            if ulp_logit < 1e-3:
                assert relerr(c.get_a_mix(), g["a_mix"][t, k], floor=1e-30) < 2e-3
            if k == 0:
                feed_ctrl_noise(c, g, t, k)
                _, states, _, _ = c.disco_forward(g["state"][t, k], g["actions"][t, k], params, want_states=True)
                assert elemerr(states, g["states_iter0"][t]) < TOL
            a_mat = g["omega_amat"][t, k]
            theta = g["theta_after"][t, k]
        theta = g["tick_theta_rolled"][t]


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_phi_update_vs_reference(golden, name):
    """a9-a11 + a8 with the reference's costs/actions injected through SVMPC.phi's log_p hook (dust_svmpc_phi)."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    kind = str(g["kernel_kind"])
    theta = g["theta0"]
    for t in range(T):
        for k in range(K):
            mu, mix = _prior_at(g, t, theta)
            c.set_theta(theta)
            c.set_prior(mu, mix)
            phi, gl, gp = c.svmpc_phi(g["costs"][t, k], g["actions"][t, k])
            assert elemerr(gp, g["grad_pri"][t, k]) < TOL, (name, t, k)
            tol = k1_tolerance(theta) if kind == "K1" else k2_tolerance(theta, c.get_bandwidths(), kind == "K2shared", int(g["da"]))
            assert elemerr(phi, g["phi"][t, k]) < tol, (name, t, k)
            theta = g["theta_after"][t, k]
        theta = g["tick_theta_rolled"][t]


def test_cloned_context_keeps_its_k2_bandwidth_settings(golden):
    """the reference's drivers deep-copy the controller inside their loops (simulations.py, particle_example.py); a copy
    must keep RBF(minimum_bw=) - host-side state set after dust_create - or it falls back to the 1e-5 clamp silently.  The clone of
    the `pend_k2_minbw` context (about half of its bandwidths clamped at 1.4) against the reference, and bit for bit against its source."""
    import copy

    name = "pend_k2_minbw"
    g = golden(name)
    src = make_ctx(g, name)
    c = copy.deepcopy(src)
    theta = g["theta0"]
    mu, mix = _prior_at(g, 0, theta)
    for x in (src, c):
        x.set_theta(theta)
        x.set_prior(mu, mix)
    phi0, _, _ = src.svmpc_phi(g["costs"][0, 0], g["actions"][0, 0])
    phi1, _, gp = c.svmpc_phi(g["costs"][0, 0], g["actions"][0, 0])
    h = c.get_bandwidths()
    assert 2 <= int((h == np.float32(g["k2_minimum_bw"])).sum()) <= h.size - 2, "the clone lost minimum_bw"
    assert np.array_equal(phi0, phi1)
    assert elemerr(phi1, g["phi"][0, 0]) < k2_tolerance(theta, h, False, int(g["da"]))
    c.close()
    src.close()


@pytest.mark.parametrize("name", K1_F64_CASES)
def test_k1_phi_vs_float64_reference(golden, name):
    """The K1 branch on the HIP path against the reference's own K1 call evaluated in float64 on the recorded inputs (no fp32
    matmul-trick noise on that side): 1e-5 element-wise, with the reference's costs / actions injected (dust_svmpc_phi)."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    for t in range(T):
        for k in range(K):
            theta = g["theta_in"][t, k]
            mu, mix = _prior_at(g, t, theta)
            c.set_theta(theta)
            c.set_prior(mu, mix)
            phi, gl, gp = c.svmpc_phi(g["costs"][t, k], g["actions"][t, k])
            assert elemerr(gl + gp, g["score"][t, k]) < TOL, (name, t, k)
            assert elemerr(phi, g["phi_f64"][t, k]) < TOL, (name, t, k)


@pytest.mark.parametrize("name", ["pend_k1_adam", "part_k1_adam"])
def test_adam_steps_vs_reference(golden, name):
    """SVMPC with the reference's class-default optimiser (torch.optim.Adam, svgd.py:115): every optimiser step of three ticks
    from the reference's own particles / noise, and the RESET of the optimiser state at each forward() (SVMPC.roll builds a new
    parameter tensor).  theta after a step is held to 1e-4 element-wise: the step direction m / sqrt(v) comes from phi, which
    inherits the softmax amplification of cost ulps (module docstring)."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    c.set_theta(g["theta0"])
    c.set_prior(g["mu0"], g["mix0"])
    c.set_a_mat(g["a_mat0"])
    for t in range(T):
        params = g["params"][t] if "params" in g else None
        c.svmpc_optimize(g["state"][t, 0], K, g["eps"][t], params)  # K Adam steps from zero state (persistent kernel where eligible)
        assert elemerr(c.get_theta(), g["theta_after"][t, K - 1]) < 1e-4, (name, t)
        a_seq, pw = c.svmpc_forward()
        assert elemerr(c.get_theta(), g["tick_theta_rolled"][t]) < 1e-4, (name, t)
    # the same three ticks through the one-call tick entry point
    c2 = make_ctx(g, name)
    c2.set_theta(g["theta0"])
    c2.set_prior(g["mu0"], g["mix0"])
    c2.set_a_mat(g["a_mat0"])
    for t in range(T):
        params = g["params"][t] if "params" in g else None
        c2.svmpc_tick(g["state"][t, 0], K, g["eps"][t], params)
        assert elemerr(c2.get_theta(), g["tick_theta_rolled"][t]) < 1e-4, (name, t)
    # which kernel served the compared ticks: the owner-computes one-launch kernel from the second tick on
    for ctx in (c, c2):
        st = ctx.tick_stats()
        assert st["tick2"] == tick2_ticks_expected(g, name) and st["replayed"] == 0, (name, st)


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_tick_chain_vs_oracle_and_reference(golden, name):
    """Whole ticks through the product entry points (optimize + forward), noise replayed.  End-to-end values inherit the
    softmax amplification of cost ulps, so they are compared (a) with the oracle run on the same chain at 2e-3 and
    (b) with the reference at the same bound; argmax / a_seq must agree exactly when the top weight is well separated."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    c.set_theta(g["theta0"])
    c.set_prior(g["mu0"], g["mix0"])
    c.set_a_mat(g["a_mat0"])
    for t in range(T):
        params = g["params"][t] if "params" in g else None
        feed_ctrl_noise(c, g, t)
        c.svmpc_optimize(g["state"][t, 0], K, g["eps"][t], params)
        th = c.get_theta()
        scale = np.abs(g["theta_after"][t, K - 1]).max()
        assert np.abs(th - g["theta_after"][t, K - 1]).max() / scale < 2e-3, (name, t)
        a_seq, pw = c.svmpc_forward()
        ref_pw = g["tick_p_weights"][t]
        srt = np.sort(ref_pw)
        if srt[-1] > 1.5 * srt[-2]:
            assert int(np.argmax(pw)) == int(np.argmax(ref_pw))
            assert np.abs(a_seq - g["tick_a_seq"][t]).max() / scale < 2e-3
        assert abs(float(pw.sum()) - 1.0) < 5e-4  # log-weights are O(1e3): one fp32 ulp there is 1e-4 relative on a weight
        # re-synchronise with the reference before the next tick so the comparison stays stage-local
        c.set_theta(g["tick_theta_rolled"][t])
    st = c.tick_stats()  # which kernel served the compared ticks: the owner-computes one-launch kernel from the second tick on
    assert st["tick2"] == tick2_ticks_expected(g, name) and st["replayed"] == 0, (name, st)


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_forward_vs_reference(golden, name):
    """a12 fed with the reference's last costs: log_l, log_p, p_weights, argmax, a_seq, roll, prior refresh."""
    g = golden(name)
    c = make_ctx(g, name)
    T, K = g["eps"].shape[:2]
    for t in range(T):
        th = g["theta_after"][t, K - 1]
        mu, mix = _prior_at(g, t, th)
        # put the device in the state the reference was in before forward(): theta, prior, last costs
        c.set_theta(g["theta0"] if K == 1 and t == 0 else (g["theta_after"][t, K - 2] if K > 1 else g["tick_theta_rolled"][t - 1]))
        params = g["params"][t, K - 1] if "params" in g else None
        # a_mat as it stood before the last iteration (it enters the costs when ctrl_penalty != 1)
        a_prev = g["omega_amat"][t, K - 2] if K > 1 else (g["a_mat0"] if t == 0 else g["omega_amat"][t - 1, K - 1])
        c.set_a_mat(a_prev)
        feed_ctrl_noise(c, g, t, K - 1)
        c.likelihood_sample(g["state"][t, K - 1], g["eps"][t, K - 1], params)
        c.set_theta(th)
        c.set_prior(mu, mix)
        a_seq, pw = c.svmpc_forward()
        ll, lp = c.get_log_weights()
        assert elemerr(ll, g["tick_log_l"][t]) < TOL
        assert elemerr(lp, g["tick_log_p"][t]) < TOL
        assert int(np.argmax(pw)) == int(np.argmax(g["tick_p_weights"][t]))
        assert relerr(pw, g["tick_p_weights"][t]) < 5e-3  # exp of O(1e3) log-weights
        assert np.array_equal(a_seq, g["tick_a_seq"][t])
        assert relerr(c.get_theta(), g["tick_theta_rolled"][t]) < 1e-6
        means, probs = c.get_prior()
        assert relerr(means, g["tick_prior_means"][t]) < 1e-6
        assert relerr(probs, g["tick_prior_probs"][t]) < 5e-3


def test_disco_mppi(golden):
    g = golden("disco_mppi")
    from dust_amd import Context

    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    temp, a_reg = float(g["temperature"]), float(g["a_reg"])
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, temperature=temp, ctrl_penalty=1.0 - a_reg / temp, sigma_a=float(g["sigma_a"]),
                alpha=1.0 / temp)
    c.set_a_mat(g["a_mat0"])
    costs, states, actions, omega = c.disco_forward(g["state"], g["actions"][0], want_states=True, around_a_mat=True)
    assert elemerr(states, g["states"]) < TOL
    assert elemerr(costs, g["costs"]) < TOL
    assert relerr(omega, g["omega"]) < 2e-4
    assert relerr(c.get_a_mat(), g["a_mat1"]) < 1e-4
    assert relerr(c.get_a_mix(), g["a_mix"]) < 2e-4
    for strat in ("argmax", "average"):
        cc = c.clone()
        nxt = cc.disco_step(strat, 2)
        assert relerr(nxt, g["step_%s_actions" % strat]) < 1e-4
        assert relerr(cc.get_a_seq(), g["step_%s_a_seq" % strat]) < 1e-4
        assert relerr(cc.get_a_mat(), g["step_%s_a_mat" % strat]) < 1e-4
    cc = c.clone()
    nxt = cc.disco_step("external", 1, g["step_external_in"])
    assert np.array_equal(nxt, g["step_external_actions"])
    assert np.array_equal(cc.get_a_seq(), g["step_external_a_seq"])
    # internally drawn noise (Philox): statistically, not bitwise, comparable - actions centred on a_mat with std sigma_a
    c2 = Context(model="pendulum", N=N, S=4096, M=1, H=H, temperature=temp, sigma_a=1.5, alpha=1.0 / temp, seed=7)
    c2.set_a_mat(g["a_mat0"])
    _, _, act, _ = c2.disco_forward(g["state"], None, want_actions=True)
    z = (act - g["a_mat0"][None]) / 1.5
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01


@pytest.mark.parametrize("name", ["mpf_pend", "mpf_part_log"])
def test_mpf(golden, name):
    from dust_amd import MpfContext
    from oracle import grid_4x4_map

    g = golden(name)
    kind = str(g["model_kind"])
    up = ("length", "mass") if kind == "pendulum" else ("mass",)
    bw, ls = float(g["bw"]), bool(int(g["log_space"]))
    m = MpfContext(g["x0"], g["obs0"], model=kind, uncertain_params=up, log_space=ls, obs_std=float(g["obs_std"]), lr=float(g["lr"]),
                   init_bw=bw, grid=grid_4x4_map() if kind == "particle" else None, mass=2.0 if kind == "particle" else 1.0)
    m.condition(g["action"], g["obs1"])
    assert elemerr(m.phi(bw), g["phi0"]) < TOL
    m2 = MpfContext(g["x0"], g["obs0"], model=kind, uncertain_params=up, log_space=ls, obs_std=float(g["obs_std"]), lr=float(g["lr"]),
                    init_bw=bw, grid=grid_4x4_map() if kind == "particle" else None, mass=2.0 if kind == "particle" else 1.0)
    gn = m2.optimize(g["action"], g["obs1"], bw, int(g["n_steps"]))
    assert elemerr(m2.get_particles(), g["x_final"]) < TOL
    assert relerr(gn, g["grad_norms"]) < TOL
    gn2 = m2.optimize(g["action2"], g["obs2"], bw, int(g["n_steps"]))
    assert elemerr(m2.get_particles(), g["x_final2"]) < TOL
    assert relerr(gn2, g["grad_norms2"]) < 2e-4  # see tests/test_oracle_golden.py::test_mpf
    assert relerr(m2.prior_log_prob(g["probe"]), g["probe_log_prob"]) < TOL
    smp = m2.prior_sample(20000, seed=3)
    means, pbw = m2.get_prior()
    assert abs(float(smp.mean()) - float(means.mean())) < 0.02 and pbw == pytest.approx(bw)


def test_mpf_control_noise_vs_reference(golden):
    """The filter over Particle(deterministic=False) (likelihoods.py:30-46 -> particle.py:145-148): one control-noise vector per SVGD
    step, shared by all filter particles; the reference's recorded draws replayed through dust_mpf_set_ctrl_noise.  All three forms
    of the optimisation kernel (single workgroup, data-polled grid, counter grid) take the per-step effective action."""
    import os

    from dust_amd import MpfContext
    from oracle import grid_4x4_map

    g = golden("mpf_part_noisy")
    bw, n = float(g["bw"]), int(g["n_steps"])
    kw = dict(model="particle", uncertain_params=("mass",), log_space=True, obs_std=float(g["obs_std"]), lr=float(g["lr"]), init_bw=bw,
              grid=grid_4x4_map(), mass=2.0, deterministic=False, noise_std=tuple(float(v) for v in g["dyn_std"]))
    m = MpfContext(g["x0"], g["obs0"], **kw)
    m.condition(g["action"], g["obs1"])
    m.set_ctrl_noise(g["phi0_noise"][None])
    assert elemerr(m.phi(bw), g["phi0"]) < TOL
    for env in ({}, {"DUST_MPF_GRID": "1"}, {"DUST_MPF_GRID": "1", "DUST_MPF_POLL": "0"}):
        old = {k: os.environ.get(k) for k in ("DUST_MPF_GRID", "DUST_MPF_POLL")}
        os.environ.update(env)
        try:
            m2 = MpfContext(g["x0"], g["obs0"], **kw)
            m2.set_ctrl_noise(np.concatenate([g["noise1"], g["noise2"]]))
            gn = m2.optimize(g["action"], g["obs1"], bw, n)
            assert elemerr(m2.get_particles(), g["x_final"]) < TOL, env
            assert relerr(gn, g["grad_norms"]) < 2e-4
            gn2 = m2.optimize(g["action2"], g["obs2"], bw, n)
            assert elemerr(m2.get_particles(), g["x_final2"]) < TOL, env
            assert relerr(gn2, g["grad_norms2"]) < 2e-4
            if env:
                assert m2.stats()["grid"] == 2 and m2.stats()["fallback"] == 0, (env, m2.stats())
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    # without the recorded draws the library's own generator takes over: a different, finite answer
    m3 = MpfContext(g["x0"], g["obs0"], **kw)
    m3.optimize(g["action"], g["obs1"], bw, n)
    x3 = m3.get_particles()
    assert np.isfinite(x3).all() and elemerr(x3, g["x_final"]) > 1e-4


def test_particle_control_noise_philox_statistics():
    """Product form of the control noise (no recorded draws): a Philox stream of its own.  With zero policy noise (eps = 0) and a free
    map the final velocity of a rollout is linear in the H control-noise draws: v_H = v_0 + dt / m * sum_t (a_t + std z_t) - mean and
    variance over the S * N * M rollouts are checked against their closed forms, and two launches draw different noise."""
    from dust_amd import Context

    N, S, M, H, std, mass, dt = 64, 64, 2, 20, (0.7, 0.3), 2.0, 0.015
    c = Context(model="particle", N=N, S=S, M=M, H=H, uncertain_params=("mass",), sampling=True, with_obstacle=False, can_crash=False,
                max_speed=1e9, max_accel=1e9, deterministic=False, noise_std=std, sigma_a=1.0, mass=mass, dt=dt)
    theta = np.full((N, H, 2), 0.5, np.float32)
    c.set_theta(theta)
    st = np.zeros(4, np.float32)
    params = np.full((M, 1), mass, np.float32)
    _, s1, _, _ = c.disco_forward(st, np.broadcast_to(theta, (S, N, H, 2)).copy(), params, want_states=True)
    _, s2, _, _ = c.disco_forward(st, np.broadcast_to(theta, (S, N, H, 2)).copy(), params, want_states=True)
    assert not np.array_equal(s1, s2), "every launch draws fresh control noise (the stream position advances)"
    v = s1[..., H, 2:4].reshape(-1, 2).astype(np.float64)
    n = v.shape[0]
    for d in range(2):
        want_mean, want_std = dt / mass * H * 0.5, dt / mass * std[d] * np.sqrt(H)
        assert abs(v[:, d].mean() - want_mean) < 5 * want_std / np.sqrt(n)
        assert abs(v[:, d].std() / want_std - 1.0) < 5 / np.sqrt(2 * n)
    # the two control channels and the dynamics samples draw independently
    assert abs(np.corrcoef(v[:, 0], v[:, 1])[0, 1]) < 5 / np.sqrt(n)
    vm = s1[..., H, 2].reshape(M, -1)
    assert abs(np.corrcoef(vm[0], vm[1])[0, 1]) < 5 / np.sqrt(vm.shape[1])


@pytest.mark.parametrize("M,obst", [(4, True), (2, False), (3, True)])
def test_particle_control_noise_inside_the_rollout_kernel(M, obst, monkeypatch):
    """Round 6: with acceleration control, device-drawn control noise and nothing but costs asked for, the regular rollout kernel draws
    the noise inside its own loops (the packed pair path: one eight-normal block per two steps of a pair; the one-sample loop: four normals
    per two steps) instead of running particle_general.hpp as a first pass (DUST_NOISE_GENERAL=1 - the path the goldens with recorded
    draws pin).  The generators' streams differ, the law does not: the costs of both paths - policy noise zero, so control noise is the
    only randomness - agree in mean and spread to sampling error, on the lean kernel (likelihood sample: pair path; M = 3 leaves a third
    sample to the one-sample loop) and on the full one (omega wanted: one-sample loop); launches draw fresh noise; without noise the
    kernel is the deterministic one."""
    from dust_amd import Context
    from oracle import grid_4x4_map

    N, S, H, std = 96, 64, 24, (0.6, 0.4)
    theta = np.full((N, H, 2), 0.4, np.float32)
    theta[:, ::3, 1] = -0.7
    st = np.array([-9.0, -9.0, 0.3, -0.2], np.float32)
    params = (2.0 + 0.05 * np.arange(M, dtype=np.float32)).reshape(M, 1)
    kw = dict(model="particle", N=N, S=S, M=M, H=H, uncertain_params=("mass",), sampling=True, with_obstacle=obst, can_crash=obst,
              grid=grid_4x4_map() if obst else None, deterministic=False, noise_std=std, sigma_a=1.0, mass=2.0, dt=0.05, seed=5)
    zeros = np.zeros((S, N, H, 2), np.float32)
    acts = np.broadcast_to(theta, (S, N, H, 2)).copy()
    out = {}
    for general in ("0", "1"):
        monkeypatch.setenv("DUST_NOISE_GENERAL", general)
        c = Context(**kw)
        c.set_theta(theta); c.set_a_mat(theta)
        lean = [c.likelihood_sample(st, zeros, params).astype(np.float64) for _ in range(6)]
        full = [c.disco_forward(st, acts, params)[0].astype(np.float64) for _ in range(6)]
        assert not np.array_equal(lean[0], lean[1]) and not np.array_equal(full[0], full[1]), "every launch draws fresh control noise"
        out[general] = (np.concatenate([x.ravel() for x in lean]), np.concatenate([x.ravel() for x in full]))
        c.close()
    for k, name in ((0, "lean kernel"), (1, "full kernel")):
        a_, b_ = out["0"][k], out["1"][k]
        n = a_.size
        assert np.isfinite(a_).all() and a_.std() > 0
        se = np.sqrt(a_.var() / n + b_.var() / n)
        assert abs(a_.mean() - b_.mean()) < 5 * se, (name, a_.mean(), b_.mean(), se)
        assert abs(a_.std() / b_.std() - 1.0) < 0.06, (name, a_.std(), b_.std())
    # lean and full kernel of the inline path against each other (different loops, different streams, one law)
    a_, b_ = out["0"]
    assert abs(a_.mean() - b_.mean()) < 5 * np.sqrt(a_.var() / a_.size + b_.var() / b_.size)


@pytest.mark.parametrize("name", ["mpf_pend_adam", "mpf_part_log_adam"])
def test_mpf_adam(golden, name):
    """MPF with the reference's class-default optimiser (torch.optim.Adam): two filter updates from the reference's own run; the
    moments and the step count persist from the first optimize() to the second (mpf.py:24), a bare phi() and a clone leave / carry
    them."""
    from dust_amd import MpfContext
    from oracle import grid_4x4_map

    g = golden(name)
    kind = str(g["model_kind"])
    up = ("length", "mass") if kind == "pendulum" else ("mass",)
    bw, ls, n = float(g["bw"]), bool(int(g["log_space"])), int(g["n_steps"])
    m = MpfContext(g["x0"], g["obs0"], model=kind, uncertain_params=up, log_space=ls, obs_std=float(g["obs_std"]), lr=float(g["lr"]),
                   init_bw=bw, grid=grid_4x4_map() if kind == "particle" else None, mass=2.0 if kind == "particle" else 1.0,
                   optimizer="Adam")
    gn = m.optimize(g["action"], g["obs1"], bw, n)
    assert elemerr(m.get_particles(), g["x_final"]) < TOL
    assert relerr(gn, g["grad_norms"]) < 2e-4
    m.phi(bw)  # takes no optimiser step
    mc = m.clone()  # carries the optimiser state
    for mm in (m, mc):
        gn2 = mm.optimize(g["action2"], g["obs2"], bw, n)
        assert elemerr(mm.get_particles(), g["x_final2"]) < TOL
        assert relerr(gn2, g["grad_norms2"]) < 2e-4


def test_collisions_and_edges(golden):
    """Occupancy lookups at edge / out-of-bounds points, through a 1-step Particle rollout whose cost isolates the map."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    g = golden("maps")
    pts = g["points"][:2048]
    coll = g["collisions"][:2048]
    grid = grid_4x4_map()
    # H=1, zero actions: cost = inst(x0) + term(x1); with w_state = w_term = w_ctrl = 0 and w_obs = 1 the cost counts
    # collisions of x0 and x1 = x0 (crashed or zero velocity), i.e. 2 * map[x0]
    o = Oracle(model="particle", N=1, S=1, M=1, H=1, uncertain_params=None, grid=grid, w_state=(0, 0, 0, 0), w_term=(0, 0, 0, 0),
               w_ctrl=(0, 0), w_obs=1.0)
    c = Context(model="particle", N=1, S=1, M=1, H=1, grid=grid, w_state=(0, 0, 0, 0), w_term=(0, 0, 0, 0), w_ctrl=(0, 0), w_obs=1.0,
                sigma_a=1.0)
    c.set_theta(np.zeros((1, 1, 2), np.float32))
    for i in list(range(0, 2048, 37)) + list(range(4000, 4010)):
        p = g["points"][i]
        st = np.array([p[0], p[1], 0.0, 0.0], np.float32)
        cost = c.likelihood_sample(st, np.zeros((1, 1, 1, 2), np.float32))
        assert float(cost[0, 0]) == 2.0 * float(g["collisions"][i]), (i, p)
        assert float(cost[0, 0]) == float(o.rollout_cost(st, np.zeros((1, 1, 1, 2), np.float32))[0, 0])
    del pts, coll


@pytest.mark.parametrize("model,N,S,M,H", [("pendulum", 256, 128, 1, 30), ("pendulum", 64, 32, 3, 7), ("particle", 128, 64, 4, 40),
                                           ("pendulum", 100, 40, 1, 7), ("pendulum", 33, 130, 2, 3), ("particle", 48, 96, 2, 9),
                                           ("pendulum", 70, 200, 1, 33)])
def test_seeded_vs_oracle(model, N, S, M, H):
    """Seeded inputs at sizes the oracle finishes in seconds: rollout/costs, score, phi (K1 and K2), forward."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    rng = np.random.default_rng(N + S)
    da = 1 if model == "pendulum" else 2
    sig = 2.0 if model == "pendulum" else 5.0
    up = None if M == 1 else (("length", "mass") if model == "pendulum" else ("mass",))
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, da))).astype(np.float32)
    eps = rng.standard_normal((S, N, H, da)).astype(np.float32)
    state = np.array([3.0, 0.0] if model == "pendulum" else [-5.2, -7.3, 4.0, 3.0], np.float32)
    params = None if up is None else rng.uniform(0.6, 1.3, (M, len(up))).astype(np.float32)
    grid = grid_4x4_map() if model == "particle" else None
    kw = dict(model=model, N=N, S=S, M=M, H=H, uncertain_params=up)
    o = Oracle(grid=grid, **kw)
    sg = np.full(da, sig, np.float32)
    actions = o.sample_actions(theta, eps, sg)
    ref_costs = o.rollout_cost(state, actions, params)
    for kernel in ("K1", "K2"):
        c = Context(grid=grid, kernel=kernel, lr=0.5, alpha=1.0 if model == "pendulum" else 1e-4, sigma_a=sig, sigma_p=sig, **kw)
        c.set_theta(theta)
        c.set_prior(mu)
        c.set_a_mat(theta)
        costs = c.likelihood_sample(state, eps, params)
        assert elemerr(costs, ref_costs) < TOL
        alpha = 1.0 if model == "pendulum" else 1e-4
        gl, gp, sc = o.score(theta, mu, np.ones(N), sg, ref_costs, actions, alpha, sg)
        phi, dgl, dgp = c.svmpc_phi(ref_costs, actions)
        assert elemerr(dgl, gl) < TOL and elemerr(dgp, gp) < TOL
        ref_phi = o.phi_k1(theta, sc) if kernel == "K1" else o.phi_k2(theta, sc)[0]
        assert elemerr(phi, ref_phi) < TOL, kernel
        if kernel == "K2":
            assert relerr(c.get_bandwidths(), o.phi_k2(theta, sc)[1]) < TOL


@pytest.mark.parametrize("model,N,S,M,H", [("pendulum", 64, 128, 1, 30), ("particle", 32, 64, 2, 20), ("pendulum", 33, 40, 3, 7)])
def test_fp16_storage_mode_vs_oracle(model, N, S, M, H):
    """BASELINE.json config 5 ("fp16 rollout / fp32 SVGD"): binary16 is a STORAGE format of the rollout's bulk data - the
    noise / actions the kernel reads (DUST_EPS_F16) and the states / actions it stores (DUST_STORE_F16) - while every
    operation stays fp32.  So the oracle fed the same binary16-rounded noise must agree to the fp32 tolerance (1e-5 relative),
    the SVGD step taken from it too, and the stored states are the fp32 states rounded once to binary16."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    rng = np.random.default_rng(5 * N + S)
    da = 1 if model == "pendulum" else 2
    sig = 2.0 if model == "pendulum" else 5.0
    up = None if M == 1 else (("length", "mass") if model == "pendulum" else ("mass",))
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, da))).astype(np.float32)
    eps16 = rng.standard_normal((S, N, H, da)).astype(np.float16)
    state = np.array([3.0, 0.0] if model == "pendulum" else [-5.2, -7.3, 4.0, 3.0], np.float32)
    params = None if up is None else rng.uniform(0.6, 1.3, (M, len(up))).astype(np.float32)
    grid = grid_4x4_map() if model == "particle" else None
    kw = dict(model=model, N=N, S=S, M=M, H=H, uncertain_params=up)
    o = Oracle(grid=grid, **kw)
    sg = np.full(da, sig, np.float32)
    actions = o.sample_actions(theta, eps16.astype(np.float32), sg)
    ref_costs, ref_states = o.rollout_cost(state, actions, params, want_states=True)
    alpha = 1.0 if model == "pendulum" else 1e-4
    c = Context(grid=grid, kernel="K1", lr=0.5, alpha=alpha, sigma_a=sig, sigma_p=sig, **kw)
    c.set_theta(theta)
    c.set_prior(mu)
    c.set_a_mat(theta)
    costs, aout = c.likelihood_sample(state, eps16, params, want_actions=True)
    assert elemerr(costs, ref_costs) < TOL and np.array_equal(aout, actions)
    # binary16 actions in, binary16 states / actions out (MultiDISCO.forward with external actions)
    act16 = actions.astype(np.float16)
    costs2, st16, a16, _ = c.disco_forward(state, act16, params, want_states=True, want_actions=True, store_f16=True)
    ref2, ref_states2 = o.rollout_cost(state, act16.astype(np.float32), params, want_states=True)
    assert st16.dtype == np.float16 and a16.dtype == np.float16
    assert elemerr(costs2, ref2) < TOL and np.array_equal(a16, act16)
    ok = np.isfinite(ref_states2) & (np.abs(ref_states2) < 6.0e4)
    d = np.abs(st16.astype(np.float32) - ref_states2)[ok]
    # one binary16 rounding (half an ulp, a whole one when the fp32 values straddle a tie) of states that agree to TOL
    assert np.all(d <= 2.0 ** -10 * np.abs(ref_states2[ok]) + TOL * np.abs(ref_states2[ok]).max())
    # one SVGD step from binary16 noise == the same step from that noise widened to fp32 on the host
    c2 = Context(grid=grid, kernel="K1", lr=0.5, alpha=alpha, sigma_a=sig, sigma_p=sig, **kw)
    for cc in (c, c2):
        cc.set_theta(theta)
        cc.set_prior(mu)
        cc.set_a_mat(theta)
    pr = None if params is None else params[None]
    c.svmpc_optimize(state, 1, eps16[None], pr)
    c2.svmpc_optimize(state, 1, eps16.astype(np.float32)[None], pr)
    assert np.array_equal(c.get_theta(), c2.get_theta())
    c.close()
    c2.close()


@pytest.mark.parametrize("world,overlap", [(2, False), (4, False), (2, True), (4, True)])
def test_sharded_equals_unsharded(golden, world, overlap):
    """Particle sharding (the multi-GPU path) on ONE GPU: `world` sharded contexts in one process, the RCCL all-gathers
    replaced by slice copies (LocalComm).  Every shard must end with the same particles as the unsharded context."""
    from dust_amd import Context
    from dust_amd.parallel import DeviceShard, LocalComm, tick

    g = golden("pend_k1")
    kw = ctx_kwargs(g)
    T, K = g["eps"].shape[:2]
    ref = Context(**kw)
    ref.set_theta(g["theta0"]); ref.set_prior(g["mu0"]); ref.set_a_mat(g["a_mat0"])
    shards = tuple(DeviceShard(kw, r, world) for r in range(world))
    for s in shards:
        s.set_state(g["theta0"], g["mu0"], g["a_mat0"])
    for t in range(T):
        ref.svmpc_optimize(g["state"][t, 0], K, g["eps"][t])
        ra, rp = ref.svmpc_forward()
        a_seq, pw = tick(shards, LocalComm(), g["state"][t, 0], K, g["eps"][t], None, want_outputs=True, overlap=overlap)
        for s in shards:
            s.sync()
        rt = ref.get_theta()
        for s in shards:
            assert relerr(s.ctx.get_theta(), rt) < (2e-6 if overlap else 1e-6), (world, t, s.rank)  # split: other merge rounding
        # (from its second tick on the unsharded context runs the owner-computes one-launch kernel, which sums over samples and keys in
        #  another order than the shards' launch-per-iteration kernels: the chosen row agrees to an ulp, not bit for bit)
        assert relerr(pw, rp) < 1e-5 and relerr(a_seq, ra) < 1e-6


def _synthetic_case(model, N, S, M, H, seed=3):
    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(seed)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    up = None if M == 1 else ("mass",)
    grid = None
    if model == "particle":
        from oracle import grid_4x4_map  # data only (the demo's occupancy grid)

        grid = grid_4x4_map()
    return da, rng, mu, th, state, up, grid


@pytest.mark.parametrize("model,N,S,M,H,kernel", [("pendulum", 256, 64, 1, 12, "K1"), ("particle", 128, 64, 4, 40, "K1"),
                                                  ("particle", 96, 32, 1, 40, "IMQ"), ("pendulum", 128, 64, 1, 10, "K2")])
def test_sharded_ticks_equal_unsharded_synthetic(model, N, S, M, H, kernel):
    """Particle sharding on ONE GPU (2 and 4 sharded contexts in one process, all-gathers as slice copies) beyond the Pendulum
    golden: Particle with D = H * da = 80 (the cfg4 shape), the IMQ kernel, and K2, whose per-dimension median bandwidths are
    GLOBAL order statistics over every rank's particles."""
    from dust_amd import Context
    from dust_amd.parallel import DeviceShard, LocalComm, tick

    da, rng, mu, th, state, up, grid = _synthetic_case(model, N, S, M, H)
    K, T = 2, 2
    eps = rng.standard_normal((T, K, S, N, H, da)).astype(np.float32)
    params = None if M == 1 else (1.0 + 0.1 * rng.standard_normal((T, K, M, 1))).astype(np.float32)
    kw = dict(model=model, N=N, S=S, M=M, H=H, kernel=kernel, lr=0.5, sigma_a=1.0, sigma_p=1.0, uncertain_params=up, seed=11)
    ref = Context(grid=grid, **kw)
    ref.set_theta(th); ref.set_prior(mu); ref.set_a_mat(th)
    outs = []
    for t in range(T):
        outs.append(ref.svmpc_tick(state, K, eps[t], None if params is None else params[t]))
    rt = ref.get_theta()
    for world in (2, 4):
        shards = tuple(DeviceShard(dict(kw, grid=grid), r, world) for r in range(world))
        for sh in shards:
            sh.set_state(th, mu, th)
        for t in range(T):
            # world 4: WITHOUT the gather of the rolled rows - every shard rolls all N rows itself (forward_finish on a sharded
            # context; what the C-side sharded tick relies on to end without a third all-gather of theta)
            a_seq, pw = tick(shards, LocalComm(), state, K, eps[t], None if params is None else params[t], want_outputs=True,
                             final_gather=(world == 2))
            assert np.array_equal(a_seq, outs[t][0]), (world, t)
            assert relerr(pw, outs[t][1]) < 1e-5
        for sh in shards:
            sh.sync()
            assert elemerr(sh.ctx.get_theta(), rt) < 2e-6, (world, sh.rank)
            sh.ctx.close()
    ref.close()


@pytest.mark.parametrize("model,N,S,M,H", [("pendulum", 256, 64, 1, 12), ("particle", 128, 64, 4, 40)])
def test_c_side_rccl_tick_world1(model, N, S, M, H):
    """The C-side sharded tick (dust_comm_init: the context's own RCCL communicator, all-gathers issued on its stream by
    libdust_amd) on a world of ONE rank, forced through the sharded code path (DUST_COMM_FORCE): every ncclAllGather runs, the
    result must equal the unsharded launch-per-iteration tick bit for bit.  (World sizes > 1 need one GPU per rank: the driver's
    multi-GPU bench; the phase order itself is covered by the LocalComm / gloo tests.)"""
    from dust_amd import Context

    da, rng, mu, th, state, up, grid = _synthetic_case(model, N, S, M, H)
    K, T = 3, 3
    eps = rng.standard_normal((T, K, S, N, H, da)).astype(np.float32)
    params = None if M == 1 else (1.0 + 0.1 * rng.standard_normal((T, K, M, 1))).astype(np.float32)
    kw = dict(model=model, N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, uncertain_params=up, grid=grid, seed=11)
    res = []
    # (third run: the particle all-gather on the main stream instead of under the next iteration's rollouts - DUST_NO_COMM_OVERLAP)
    for env, sharded in (({"DUST_NO_PERSIST": "1"}, False), ({"DUST_COMM_FORCE": "1"}, True),
                         ({"DUST_COMM_FORCE": "1", "DUST_NO_COMM_OVERLAP": "1"}, True)):
        saved = {k: os.environ.pop(k, None) for k in ("DUST_NO_PERSIST", "DUST_COMM_FORCE", "DUST_NO_COMM_OVERLAP")}
        os.environ.update(env)
        try:
            c = Context(shard_offset=0, shard_size=N, **kw) if sharded else Context(**kw)
            if sharded:
                c.comm_init(Context.comm_unique_id(), 0, 1)
            c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
            outs = [c.svmpc_tick(state, K, eps[t], None if params is None else params[t]) for t in range(T)]
            res.append((c.get_theta(), outs))
            c.close()
        finally:
            for k in ("DUST_NO_PERSIST", "DUST_COMM_FORCE", "DUST_NO_COMM_OVERLAP"):
                os.environ.pop(k, None)
                if saved[k] is not None:
                    os.environ[k] = saved[k]
    (t0, o0), (t1, o1), (t2, o2) = res
    # without the overlap the sharded tick runs the unsharded tick's kernels: the same bits.  With it, iterations 2.. form the score
    # in prior_finish_kernel instead of the rollout kernel's merge epilogue (another order of the slice combine: one ulp)
    assert np.array_equal(t0, t2) and elemerr(t1, t0) < 1e-5
    for (a0, p0), (a1, p1), (a2, p2) in zip(o0, o1, o2):
        assert np.array_equal(a0, a2) and relerr(p2, p0) < 1e-5
        assert elemerr(a1, a0) < 1e-5 and relerr(p1, p0) < 1e-4


@pytest.mark.parametrize("model,N,S,M,H", [("pendulum", 1024, 128, 1, 30), ("pendulum", 512, 64, 1, 12), ("particle", 256, 64, 4, 20),
                                           ("pendulum", 100, 40, 1, 7), ("pendulum", 33, 130, 3, 3), ("particle", 64, 96, 2, 9),
                                           ("pendulum", 96, 256, 1, 33), ("particle", 40, 30, 1, 31)])
@pytest.mark.parametrize("kernel,optimizer", [("K1", "SGD"), ("IMQ", "Adam")])
def test_fused_launches_equal_unfused_bitwise(model, N, S, M, H, kernel, optimizer):
    """Every in-launch hand-off form against the same bodies run as separate kernels (per-kernel profiling on): the one-launch SVGD
    iteration and the two fused launches (fused.hpp) must give the same BITS, tick after tick: a hand-off that lets a consumer run
    early (or a producer overwrite an input another workgroup still reads) shows up here.  (The owner-computes tick, tick2.hpp, sums
    over keys and samples in another order: it is switched off here and held to these paths and to the oracle at a tolerance in
    tests/test_gpu_tick2.py.  The tiled one-launch tick of rounds 2-5, persist.hpp, is retired.)"""
    from dust_amd import Context

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(3)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    up = None if M == 1 else ("mass",)
    params = None if M == 1 else (1.0 + 0.1 * rng.standard_normal((3, M, 1))).astype(np.float32)
    grid = None
    if model == "particle":
        from oracle import grid_4x4_map  # data only (the demo's occupancy grid)

        grid = grid_4x4_map()

    def run(env, unfused, n_ticks):
        saved = {k: os.environ.pop(k, None) for k in ("DUST_NO_PERSIST", "DUST_NO_TICK2")}
        os.environ.update(env)
        os.environ["DUST_NO_TICK2"] = "1"
        try:
            c = Context(model=model, N=N, S=S, M=M, H=H, kernel=kernel, lr=0.5 if optimizer == "SGD" else 0.05, optimizer=optimizer,
                        sigma_a=1.0, sigma_p=1.0, uncertain_params=up, grid=grid, seed=11)
            c.set_theta(th)
            c.set_prior(mu)
            c.set_a_mat(th)
            c.profile(unfused)
            for _ in range(n_ticks - 2):
                c.svmpc_tick(state, 3, params=params, want_outputs=False)
            a_seq, pw = c.svmpc_tick(state, 3, params=params, want_outputs=True)
            c.svmpc_tick(state, 2, params=None if params is None else params[:2], want_outputs=False)  # even count: in-place roll
            c.sync()
            out = (c.get_theta(), c.get_score(), c.get_a_mat(), a_seq, pw)
            c.close()
            return out
        finally:
            for k in ("DUST_NO_PERSIST", "DUST_NO_TICK2"):
                os.environ.pop(k, None)
                if saved[k] is not None:
                    os.environ[k] = saved[k]

    ref = run({}, True, 6)
    for env in ({}, {"DUST_NO_PERSIST": "1"}):
        got = run(env, False, 6)
        for a, b, what in zip(got, ref, ("theta", "score", "a_mat", "a_seq", "p_weights")):
            assert np.array_equal(a, b), (env, what)


@pytest.mark.parametrize("N,H,ell", [(256, 30, 1.0), (1024, 30, 0.7), (96, 40, 2.5)])
def test_imq_phi_vs_oracle(N, H, ell):
    """IMQ kernel k = (1 + |x-y|^2/l^2)^(-1/2) (new with the north star, pinned only by the oracle's closed form)."""
    from dust_amd import Context
    from oracle import Oracle

    rng = np.random.default_rng(N)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + 0.5 * rng.standard_normal((N, H, 1))).astype(np.float32)
    S = 16
    costs = (50.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, 1))).astype(np.float32)
    o = Oracle(model="pendulum", N=N, S=S, M=1, H=H)
    sg = np.ones(1, np.float32)
    gl, gp, sc = o.score(theta, mu, np.ones(N), sg, costs, actions, 1.0, sg)
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="IMQ", imq_ell=ell, lr=0.5, sigma_a=1.0, sigma_p=1.0)
    c.set_theta(theta)
    c.set_prior(mu)
    c.set_a_mat(theta)
    phi, dgl, dgp = c.svmpc_phi(costs, actions)
    assert elemerr(dgl, gl) < TOL and elemerr(dgp, gp) < TOL
    assert elemerr(phi, o.phi_imq(theta, sc, ell)) < TOL
    c.close()


@pytest.mark.parametrize("N,optimizer", [(1024, "SGD"), (1000, "Adam"), (200, "SGD"), (70, "Adam")])
def test_k2_launch_forms_agree(N, optimizer, monkeypatch):
    """Round 6, bandwidth.hpp: by default the per-dimension bandwidths ride in the prior + rollout launch as a 256-lane role
    (k2_bandwidth256) and phi reads the row-major particles, updating into the other particle buffer - no transposed copy;
    DUST_K2_FORM=0 keeps the launches of rounds 1-5 (transpose, 1024-lane bandwidth kernel, phi in place).  The bandwidths are exact order
    statistics either way (test_k2_bandwidth_is_the_exact_order_statistic runs on the default) and phi is the same kernel on the same
    values: the particles are BIT-IDENTICAL over ticks of several iterations - ragged particle counts, both optimisers, odd and even
    iteration counts (the particle buffers ping-pong in the default form)."""
    from dust_amd import Context

    H, S = 7, 16
    rng = np.random.default_rng(N)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + 1.5 * rng.standard_normal((N, H, 1))).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    out = {}
    for form in ("0", "2", "pinned"):
        monkeypatch.setenv("DUST_K2_FORM", "2" if form == "pinned" else form)
        c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K2", lr=0.5 if optimizer == "SGD" else 0.05, sigma_a=1.0, sigma_p=1.0,
                    optimizer=optimizer, seed=11)
        c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
        if form == "pinned":  # the particles' address handed out (in-place collectives): one buffer, phi through a transposed copy again
            import ctypes as C
            from dust_amd import _lib as L

            th, sc, nb = L.VP(), L.VP(), C.c_size_t(0)
            L.check(L.load().dust_gather_buffers(c._h, C.byref(th), C.byref(sc), C.byref(nb)))
        hs = []
        for it in (3, 3, 3, 3, 2, 2, 2):  # (the third tick of a shape on is a replayed capture)
            c.svmpc_tick(state, it)
            hs.append(c.get_bandwidths().copy())
        out[form] = (c.get_theta(), np.array(hs))
        c.close()
    for form in ("2", "pinned"):
        assert np.array_equal(out[form][1], out["0"][1]), "bandwidths differ (%s)" % form
        assert np.array_equal(out[form][0], out["0"][0]), (form, elemerr(out[form][0], out["0"][0]))


@pytest.mark.parametrize("N", [8, 64, 260, 1000, 1024, 2048, 3000])
def test_k2_bandwidth_is_the_exact_order_statistic(N):
    """h_c = median(pairwise squared distances) / log(N+1) with torch.median's lower-middle rule over all N^2 entries
    (base_kernels.py:53-77): the GPU selects it by bisection on the float's bit pattern - it must be the exact value, for
    both bandwidth kernels (N <= 1024 bracketed search, N > 1024 plain)."""
    from dust_amd import Context

    H = 3
    rng = np.random.default_rng(N)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + rng.standard_normal((N, H, 1))).astype(np.float32)
    theta[: N // 7, 1, 0] = theta[0, 1, 0]  # ties and zero distances in one dimension
    if N == 260:
        theta[:, 2, 0] = 1.25  # a constant column: every distance is zero, the bandwidth is the clamp
    c = Context(model="pendulum", N=N, S=8, M=1, H=H, kernel="K2", lr=0.0, sigma_a=1.0, sigma_p=1.0)
    c.set_theta(theta)
    c.set_prior(mu)
    c.set_a_mat(theta)
    c.svmpc_optimize(np.array([3.0, 0.0], np.float32), 1)
    h = c.get_bandwidths()

    def exact(X):
        out = []
        for d in range(H):
            x = X[:, d]
            pw = (x[:, None] - x[None, :]) ** 2  # fp32, as the sorted-coordinate kernel forms it
            v = np.partition(pw.ravel(), (N * N - 1) // 2)[(N * N - 1) // 2]
            out.append(max(np.float32(v) / np.float32(np.log(N + 1.0)), np.float32(1e-5)))
        return out

    ref = exact(theta.reshape(N, H))
    for d in range(H):
        assert np.float32(h[d]) == np.float32(ref[d]), (d, h[d], ref[d])
    # later calls start the bisection from the previous bandwidth (warm start, N <= 1024): nudged particles (the bracket holds),
    # the same particles again (the answer sits ON the previous value), and a jump (the bracket misses: full range)
    for scale, shift in ((1.002, 0.0), (1.0, 0.0), (1.7, 0.3)):
        theta = (theta * np.float32(scale) + np.float32(shift) * rng.standard_normal(theta.shape).astype(np.float32)).astype(np.float32)
        c.set_theta(theta)
        c.svmpc_optimize(np.array([3.0, 0.0], np.float32), 1)
        h = c.get_bandwidths()
        ref = exact(theta.reshape(N, H))
        for d in range(H):
            assert np.float32(h[d]) == np.float32(ref[d]), (scale, d, h[d], ref[d])
    c.close()


@pytest.mark.parametrize("model,N,H,kernel", [("pendulum", 2048, 30, "K1"), ("pendulum", 2200, 17, "IMQ"), ("particle", 2048, 20, "K1"),
                                              ("particle", 2048, 40, "K1"), ("pendulum", 4096, 30, "K1")])
def test_large_key_set_pairwise_vs_oracle(model, N, H, kernel, monkeypatch):
    """N >= 2048 takes the register-blocked pairwise kernel (pairwise_big.hpp): prior score, log p and phi against the
    oracle for both padded widths (D = 30 / 17 -> 32, 40 -> 64), D = 80 (which stays on the 32 x 64 kernel), ragged N, and
    log p through a forward pass."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")  # (N D <= 65536 runs the small-set launches by default: this test is about the large-set kernels)
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(N + H)
    S = 8
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + 0.4 * rng.standard_normal((N, H, da))).astype(np.float32)
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, da))).astype(np.float32)
    grid = grid_4x4_map() if model == "particle" else None
    o = Oracle(model=model, N=N, S=S, M=1, H=H, grid=grid)
    sg = np.full(da, 1.5, np.float32)
    sp = np.array([1.5, 0.8], np.float32)[:da]
    gl, gp, sc = o.score(theta, mu, np.ones(N), sp, costs, actions, 1.0, sg)
    c = Context(model=model, N=N, S=S, M=1, H=H, kernel=kernel, imq_ell=0.9, lr=0.5, sigma_a=sg, sigma_p=sp, grid=grid)
    c.set_theta(theta)
    c.set_prior(mu)
    c.set_a_mat(theta)
    phi, dgl, dgp = c.svmpc_phi(costs, actions)
    assert elemerr(dgp, gp) < TOL
    ref = o.phi_k1(theta, sc) if kernel == "K1" else o.phi_imq(theta, sc, 0.9)
    assert elemerr(phi, ref) < (k1_tolerance(theta) if kernel == "K1" else TOL)
    # log p(theta) of the forward pass (the prior pass without the gradient)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    c.svmpc_optimize(state, 1)
    th1 = c.get_theta()
    c.svmpc_forward()
    ll, lp = c.get_log_weights()
    fw = o.forward(c.get_costs(), th1, mu, np.ones(N), sp, 1.0)
    assert relerr(lp, fw["log_p"]) < 1e-5
    c.close()


@pytest.mark.parametrize("model,N,S,H,K", [("pendulum", 64, 32, 15, 3), ("particle", 32, 64, 20, 2)])
def test_philox_tick_replayed_through_oracle(model, N, S, H, K):
    """The product tick draws its policy noise on the device (Philox, in registers - the noise never exists in HBM).  Here the
    SAME draws are fetched as actions through the stand-alone sample entry point of a second context walked through the same
    (tick, iteration) stream positions, replayed through the CPU oracle for one whole tick (rollouts, costs, score, K1 phi, SGD,
    forward), and the result is compared with the particles the persistent one-launch tick produced from them on the device."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    da = 1 if model == "pendulum" else 2
    sig = 2.0 if model == "pendulum" else 5.0
    lr = 0.5 if model == "pendulum" else 20.0
    alpha = 1.0 if model == "pendulum" else 1e-4
    rng = np.random.default_rng(17)
    mu = rng.standard_normal((N, H, da)).astype(np.float32)
    theta = (mu + 0.5 * rng.standard_normal((N, H, da))).astype(np.float32)
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    grid = grid_4x4_map() if model == "particle" else None
    kw = dict(model=model, N=N, S=S, M=1, H=H, kernel="K1", lr=lr, alpha=alpha, sigma_a=sig, sigma_p=sig, grid=grid, seed=4242)
    prod = Context(**kw)
    prod.set_theta(theta); prod.set_prior(mu); prod.set_a_mat(theta)
    a_seq, pw = prod.svmpc_tick(state, K)  # device Philox noise, one persistent launch
    th_prod = prod.get_theta()
    # replay: a second context hands out the draws of stream positions (tick 0, iteration k) as actions around the ORACLE's theta
    tap = Context(**kw)
    tap.set_prior(mu); tap.set_a_mat(theta)
    o = Oracle(model=model, N=N, S=S, M=1, H=H, grid=grid)
    sg = np.full(da, sig, np.float32)
    th, mix = theta.copy(), np.ones(N, np.float32)
    for k in range(K):
        tap.set_theta(th)
        costs_dev, actions = tap.likelihood_sample(state, None, None, want_actions=True)  # Philox draw k of tick 0
        eps = (actions - th[None]) / sig
        assert abs(float(eps.mean())) < 0.05 and abs(float(eps.std()) - 1.0) < 0.05
        costs = o.rollout_cost(state, actions)
        assert elemerr(costs_dev, costs) < TOL
        _, _, sc = o.score(th, mu, mix, sg, costs, actions, alpha, sg)
        th = o.sgd(th, o.phi_k1(th, sc), lr)
    r = o.forward(costs, th, mu, mix, sg, alpha)
    # whole-tick chain: softmax amplification of cost ulps (module docstring) - 2e-3, as the golden chain test
    scale = np.abs(r["theta"]).max()
    assert np.abs(th_prod - r["theta"]).max() / scale < 2e-3
    srt = np.sort(r["p_weights"])
    if srt[-1] > 1.5 * srt[-2]:
        assert int(np.argmax(pw)) == r["i_star"]
        assert np.abs(a_seq - r["a_seq"]).max() / scale < 2e-3


def test_cfg5_shaped_dual_loop_vs_oracle():
    """BASELINE.json configs[4] in one loop, at reduced N: Pendulum dual inference - control SVGD with the IMQ kernel over
    binary16-STORED policy noise (fp32 arithmetic), M = 8 dynamics samples drawn per tick from the GMM over an M_p = 256
    dynamics-particle filter, then 20 MPF steps on the new observation - against the CPU oracle fed the same draws."""
    from dust_amd import Context
    from dust_amd.backend import MpfContext
    from oracle import Oracle

    N, S, H, M, Mp, K, T = 96, 64, 30, 8, 256, 2, 2
    rng = np.random.default_rng(55)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, 1))).astype(np.float32)
    x0 = rng.uniform(0.6, 1.3, (Mp, 2)).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    ell, lr, sig = 1.5, 0.5, 2.0
    c = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel="IMQ", imq_ell=ell, lr=lr, sigma_a=sig, sigma_p=sig,
                uncertain_params=("length", "mass"), seed=3)
    c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
    # (filter step size / bandwidth chosen so that 20 SVGD steps of 256 particles are a contraction: with the demo's lr = 1e-3 and
    # bw = 0.08 the 256-particle system is chaotic - device and oracle agree to 1e-7 after one step and drift apart exponentially)
    mlr, mbw = 1e-4, 0.2
    mpf = MpfContext(x0, state, model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=mlr, init_bw=mbw)
    o = Oracle(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=("length", "mass"))
    sg = np.full(1, sig, np.float32)
    th, m_, mix = theta.copy(), mu.copy(), np.ones(N, np.float32)
    xo, pmo, pbw, obs_prev = x0.copy(), x0.copy(), mbw, state.copy()
    for t in range(T):
        params = np.stack([mpf.prior_sample(M, seed=100 * t + k) for k in range(K)])  # [K][M][2] from the filter's GMM
        assert np.all(np.isfinite(params))
        eps16 = rng.standard_normal((K, S, N, H, 1)).astype(np.float16)
        a_seq, pw = c.svmpc_tick(state, K, eps16, params)
        for k in range(K):  # oracle: the same binary16-rounded noise widened on the host
            actions = o.sample_actions(th, eps16[k].astype(np.float32), sg)
            costs = o.rollout_cost(state, actions, params[k])
            mu_k = m_ if t == 0 else th  # from the second tick on the prior means ARE the current particles (svgd.py:87)
            _, _, sc = o.score(th, mu_k, mix, sg, costs, actions, 1.0, sg)
            th = o.sgd(th, o.phi_imq(th, sc, ell), lr)
        r = o.forward(costs, th, m_ if t == 0 else th, mix, sg, 1.0)
        assert np.abs(c.get_theta() - r["theta"]).max() / np.abs(r["theta"]).max() < 2e-3, t
        th, m_, mix = r["theta"], r["mu"], r["mix"]
        c.set_theta(th)  # re-synchronise (stage-local comparison), keep the device's prior refresh
        # plant + dynamics filter (simulations.py:129-138)
        action = np.clip(a_seq[0], -2.0, 2.0)
        from dust_amd.models import PendulumModel
        import torch

        plant = PendulumModel(g=10.0, length=0.9, mass=1.1)
        new_obs = plant.step(torch.tensor(obs_prev)[None], torch.tensor(action).reshape(1, 1))[0].numpy()
        mpf.optimize(action, new_obs, mbw, 20)
        xo, pmo, pbw, _ = o.mpf_optimize(xo, pmo, pbw, obs_prev, action, new_obs, 0.1, False, mbw, mlr, 20)
        assert elemerr(mpf.get_particles(), xo) < 2e-5, t
        obs_prev = new_obs
        state = new_obs.astype(np.float32)


def test_cfg5_shaped_sharded_equals_unsharded():
    """The control half of BASELINE.json configs[4] (Pendulum, IMQ kernel, M = 8 sampled dynamics per iteration, binary16-stored
    noise) SHARDED over 2 and 4 contexts against the unsharded context: the unsharded tick reads the binary16 noise, the shards the
    same values widened on the host (storage format only: the steps are identical), all-gathers as slice copies."""
    from dust_amd import Context
    from dust_amd.parallel import DeviceShard, LocalComm, tick

    N, S, H, M, K, T = 128, 64, 30, 8, 2, 2
    rng = np.random.default_rng(56)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, 1))).astype(np.float32)
    state = np.array([3.0, 0.0], np.float32)
    kw = dict(model="pendulum", N=N, S=S, M=M, H=H, kernel="IMQ", imq_ell=1.5, lr=0.5, sigma_a=2.0, sigma_p=2.0,
              uncertain_params=("length", "mass"), seed=3)
    eps16 = rng.standard_normal((T, K, S, N, H, 1)).astype(np.float16)
    params = rng.uniform(0.7, 1.3, (T, K, M, 2)).astype(np.float32)
    ref = Context(**kw)
    ref.set_theta(theta); ref.set_prior(mu); ref.set_a_mat(theta)
    outs = [ref.svmpc_tick(state, K, eps16[t], params[t]) for t in range(T)]
    rt = ref.get_theta()
    for world in (2, 4):
        shards = tuple(DeviceShard(dict(kw), r, world) for r in range(world))
        for sh in shards:
            sh.set_state(theta, mu, theta)
        for t in range(T):
            a_seq, pw = tick(shards, LocalComm(), state, K, eps16[t].astype(np.float32), params[t], want_outputs=True)
            assert np.array_equal(a_seq, outs[t][0]), (world, t)
            assert relerr(pw, outs[t][1]) < 1e-5
        for sh in shards:
            sh.sync()
            assert elemerr(sh.ctx.get_theta(), rt) < 2e-6, (world, sh.rank)
            sh.ctx.close()
    ref.close()


@pytest.mark.parametrize("model,N,H,kernel,spread", [
    ("particle", 2048, 40, "K1", 0.25), ("pendulum", 2304, 30, "K1", 0.25), ("particle", 2100, 20, "K1", 0.25),
    ("particle", 2048, 40, "IMQ", 0.25), ("pendulum", 2048, 17, "IMQ", 0.25),
    ("particle", 16384, 40, "K1", 0.25),  # the cfg4 shape itself: N = 16384, D = 80
    ("particle", 2048, 40, "K1", 1.2),    # far-apart particles: almost every Stein kernel value underflows to 0
    ("pendulum", 4096, 30, "K1", 2.0),
    ("particle", 2048, 40, "K1", 0.02)])  # nearly collapsed set: every kernel value ~ 1
def test_fused_large_pairwise_vs_oracle(model, N, H, kernel, spread, monkeypatch):
    """Prior means aliasing theta + N >= 2048: ONE distance pass serves the prior score, the Stein repulsion and the Gram matrix,
    then Gram x score runs as a GEMM (pairwise_fused.hpp) - D = 80 / 30 / 40 / 17 (tile widths 80 / 32 / 64 / 32), ragged N,
    non-uniform mixture weights.  grad_pri and phi against the oracle, and against the two unfused passes (DUST_PAIR_FUSED=0)."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")  # (N D <= 65536 runs the small-set launches by default: this test is about the large-set kernels)
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(3 * N + H)
    S = 8
    theta = (spread * rng.standard_normal((N, H, da))).astype(np.float32)  # (0.25: close enough for the mixture weights to overlap)
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, da))).astype(np.float32)
    mixw = rng.random(N).astype(np.float32) + 0.05
    mixw /= mixw.sum()
    grid = grid_4x4_map() if model == "particle" else None
    o = Oracle(model=model, N=N, S=S, M=1, H=H, grid=grid)
    sg = np.full(da, 1.5, np.float32)
    sp = np.array([1.5, 0.8], np.float32)[:da]
    gl, gp, sc = o.score(theta, theta, mixw, sp, costs, actions, 1.0, sg)
    ref = o.phi_k1(theta, sc) if kernel == "K1" else o.phi_imq(theta, sc, 0.9)
    got = {}
    for fused in ("1", "0"):
        os.environ["DUST_PAIR_FUSED"] = fused
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel=kernel, imq_ell=0.9, lr=0.5, sigma_a=sg, sigma_p=sp, grid=grid,
                        weighted_prior=True)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)  # the means alias theta from here on (svmpc.py:160-170)
            got[fused] = c.svmpc_phi(costs, actions)
            c.close()
        finally:
            os.environ.pop("DUST_PAIR_FUSED", None)
    phi, dgl, dgp = got["1"]
    # sparse cases: the surviving weights are exp(-100) and the like, where the bare v_exp_f32's relative error |x| 2^-24 shows
    tol_p = TOL if spread < 1.0 else 4e-5
    # (a spread set's grad_pri consists of nothing but terms of e^-60 and below, which pairwise_far.hpp leaves out: held to
    #  1e-3 of the scale of the score it is added to, or to its own RMS where that is larger)
    rms = lambda v: float(np.sqrt(np.mean(np.float64(v) ** 2)))
    floor_p = max(rms(gp), 1e-3 * rms(sc))
    assert elemerr(dgp, gp, floor=floor_p) < tol_p
    assert elemerr(phi, ref) < (k1_tolerance(theta) if kernel == "K1" else TOL)
    phi0, _, dgp0 = got["0"]
    assert elemerr(dgp, dgp0, floor=floor_p) < tol_p and elemerr(phi, phi0) < TOL
    if N >= 16384:  # run-to-run determinism at the size where two workgroups share every CU for thousands of chunks: an intra-
        os.environ["DUST_PAIR_FUSED"] = "1"  # workgroup race (seen once in a variant of this kernel) shows up as a few differing rows
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel=kernel, imq_ell=0.9, lr=0.5, sigma_a=sg, sigma_p=sp, grid=grid,
                        weighted_prior=True)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)
            for _ in range(3):
                phi2, _, dgp2 = c.svmpc_phi(costs, actions)
                assert np.array_equal(phi2, phi) and np.array_equal(dgp2, dgp)
            c.close()
        finally:
            os.environ.pop("DUST_PAIR_FUSED", None)


@pytest.mark.parametrize("N,H", [(2254, 56), (2986, 50), (2304, 30)])
def test_fused_pairwise_is_deterministic(N, H):
    """Twelve launches of the fused pair on the same nearly collapsed Pendulum set: bit-identical phi and grad_pri.  These are the
    shapes (64-wide tiles, ragged N) at which one launch in ten used to read four stale key rows: `s_barrier` does not wait for
    the issuing wave's queued LDS writes, and the compiler had left the `s_waitcnt lgkmcnt(0)` out of the barrier at the head of
    the chunk loop (common.hpp wg_sync; tools/fused_race.hip runs the kernel alone)."""
    from dust_amd import Context

    rng = np.random.default_rng(N + H)
    S = 4
    theta = (0.05 * rng.standard_normal((N, H, 1))).astype(np.float32)
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, 1))).astype(np.float32)
    mixw = rng.random(N).astype(np.float32) + 0.05
    os.environ["DUST_PAIR_FUSED"] = "1"
    try:
        c = Context(model="pendulum", N=N, S=S, M=1, H=H, kernel="K1", lr=0.0, sigma_a=1.5, sigma_p=1.5, weighted_prior=True, seed=5)
        c.set_theta(theta)
        c.set_prior(theta)
        c.set_a_mat(theta)
        c.svmpc_update_prior(mixw)
        phi, _, gp = c.svmpc_phi(costs, actions)
        for rep in range(11):
            phi2, _, gp2 = c.svmpc_phi(costs, actions)
            assert np.array_equal(phi2, phi) and np.array_equal(gp2, gp), rep
        c.close()
    finally:
        os.environ.pop("DUST_PAIR_FUSED", None)


def test_sharded_large_set_takes_fused_pairwise():
    """A rank of a sharded run with >= 512 local particles and N >= 2048 keys takes the fused large-set pairwise launches too
    (its Gram matrix is [n_local][N]): 2 and 4 shards in one process against the unsharded context, Particle D = 80."""
    from dust_amd import Context
    from dust_amd.parallel import DeviceShard, LocalComm, tick

    model, N, S, M, H = "particle", 2048, 8, 1, 40
    da, rng, mu, th, state, up, grid = _synthetic_case(model, N, S, M, H)
    K, T = 2, 2
    eps = rng.standard_normal((T, K, S, N, H, da)).astype(np.float32)
    kw = dict(model=model, N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=1.0, sigma_p=1.0, uncertain_params=up, seed=11)
    ref = Context(grid=grid, **kw)
    ref.set_theta(th); ref.set_prior(mu); ref.set_a_mat(th)
    outs = [ref.svmpc_tick(state, K, eps[t]) for t in range(T)]
    rt = ref.get_theta()
    for world in (2, 4):
        shards = tuple(DeviceShard(dict(kw, grid=grid), r, world) for r in range(world))
        for sh in shards:
            sh.set_state(th, mu, th)
        for t in range(T):
            a_seq, pw = tick(shards, LocalComm(), state, K, eps[t], None, want_outputs=True)
            assert np.array_equal(a_seq, outs[t][0]), (world, t)
            assert relerr(pw, outs[t][1]) < 1e-5
        for sh in shards:
            sh.sync()
            assert elemerr(sh.ctx.get_theta(), rt) < 1e-5, (world, sh.rank)
            sh.ctx.close()
    ref.close()


def test_cfg4_rank_of_eight_runs_the_large_set_kernels():
    """one rank's share of BASELINE configs[3] on 8 GPUs (N = 16384 Particle particles, 2 048 of them local) - the
    shape whose last round-3 profile showed a 4x slower tick.  That reading was one 44 ms host-side gap inside a 20-tick timed loop
    (tools/shard_time.py, profiles/round4_shard_time.txt); this test pins what the rank's tick consists of instead: in profiling mode
    (one event pair per kernel) the passes of a whole tick must stay where the fused large-set kernels put them - against 1.1 ms and
    1.0 ms for the two pairwise passes alone through the small-set kernels."""
    from dust_amd.parallel import DeviceShard
    from oracle import grid_4x4_map

    N, S, M, H = 16384, 64, 4, 40
    rng = np.random.default_rng(0)
    mu = rng.standard_normal((N, H, 2)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, 2))).astype(np.float32)
    cfg = dict(model="particle", N=N, S=S, M=M, H=H, kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0, uncertain_params=("mass",),
               grid=grid_4x4_map(), seed=3)
    sh = DeviceShard(cfg, 0, 8, use_torch_stream=False)
    sh.set_state(th, th)
    state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    params = (1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)

    def one_tick():
        sh.local_score(state, params=params)
        sh.apply_phi()
        sh.forward_local()
        sh.forward_finish()

    one_tick()  # the forward aliases the prior means to the particles: from here on the fused pass serves prior + Stein
    sh.ctx.profile(True)
    for _ in range(4):
        one_tick()
    sh.ctx.sync()
    pk = {k: 1e3 * ms / n for k, (ms, n) in sh.ctx.profile_get().items()}
    sh.ctx.close()
    assert np.isfinite(list(pk.values())).all()
    # measured: rollouts 89, fused pairwise pass 145, Gram x score 58, update 16, finalize + roll 17 us (profiles/round4_shard_time.txt)
    assert pk["pairwise_kernel<PRIOR>"] < 450 and pk["pairwise_kernel<STEIN>"] < 200 and sum(pk.values()) < 1000, pk


def test_particle_m64_small_n_vs_oracle():
    """BASELINE configs[2]'s dynamics-sample count (M = 64, S = 64, H = 40: the lane-group split of the M loop at its real trip
    counts) at a particle count the oracle finishes in seconds: every cost, then score and phi, element-wise."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    N, S, M, H = 64, 64, 64, 40
    rng = np.random.default_rng(64)
    mu = rng.standard_normal((N, H, 2)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, 2))).astype(np.float32)
    eps = rng.standard_normal((S, N, H, 2)).astype(np.float32)
    state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    params = (1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
    grid = grid_4x4_map()
    kw = dict(model="particle", N=N, S=S, M=M, H=H, uncertain_params=("mass",))
    o = Oracle(grid=grid, **kw)
    sg = np.full(2, 5.0, np.float32)
    actions = o.sample_actions(theta, eps, sg)
    ref_costs = o.rollout_cost(state, actions, params)
    c = Context(grid=grid, kernel="K1", lr=0.5, alpha=1e-4, sigma_a=5.0, sigma_p=5.0, **kw)
    c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
    costs = c.likelihood_sample(state, eps, params)
    assert elemerr(costs, ref_costs) < TOL
    gl, gp, sc = o.score(theta, mu, np.ones(N), sg, ref_costs, actions, 1e-4, sg)
    phi, dgl, dgp = c.svmpc_phi(ref_costs, actions)
    assert elemerr(dgl, gl) < TOL and elemerr(dgp, gp) < TOL and elemerr(phi, o.phi_k1(theta, sc)) < TOL
    c.close()


@pytest.mark.parametrize("name,N,S,M,H", [("cfg3", 4096, 64, 64, 40), ("cfg4", 16384, 64, 4, 40)])
def test_full_size_rollouts_sampled_vs_oracle(name, N, S, M, H):
    """BASELINE configs[2] / configs[3] at their REAL sizes (the shapes bench.py times): the costs of 64 randomly chosen particles
    of the full launch against the oracle.  A particle's rollouts depend only on its own theta row, its own noise column, the M
    dynamics samples and the plant state (disco.py:139-209 with 2-D parameter samples), so the oracle is run on the 64-particle
    subset with the same inputs; every one of the 64 x S costs is compared element-wise."""
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    rng = np.random.default_rng(N)
    mu = rng.standard_normal((N, H, 2)).astype(np.float32)
    theta = (mu + 0.3 * rng.standard_normal((N, H, 2))).astype(np.float32)
    eps = rng.standard_normal((S, N, H, 2)).astype(np.float32)
    state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    params = (1.0 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
    grid = grid_4x4_map()
    c = Context(model="particle", N=N, S=S, M=M, H=H, uncertain_params=("mass",), grid=grid, kernel="K1", lr=0.5, alpha=1e-4,
                sigma_a=5.0, sigma_p=5.0)
    c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
    costs = c.likelihood_sample(state, eps, params)  # [S][N], the full launch
    idx = np.sort(rng.choice(N, 64, replace=False))
    o = Oracle(model="particle", N=64, S=S, M=M, H=H, uncertain_params=("mass",), grid=grid)
    sg = np.full(2, 5.0, np.float32)
    actions = o.sample_actions(theta[idx], np.ascontiguousarray(eps[:, idx]), sg)
    ref = o.rollout_cost(state, actions, params)
    assert elemerr(costs[:, idx], ref) < TOL, name
    c.close()


def test_cfg5_full_size_sampled_vs_oracle():
    """BASELINE configs[4] at its REAL size (Pendulum N = 2048, S = 128, M = 8 sampled (length, mass), H = 30, binary16 rollout noise,
    IMQ kernel - the shape tools/configs_bench.py times; VERDICT r3: cfg5 was only compared 'shaped' at N = 96 / 128).  (a) The costs
    of 64 randomly chosen particles of the full launch against the oracle fed the same binary16 noise widened to fp32 (a particle's
    rollouts depend on its own row, its own noise column, the M dynamics samples and the plant state: disco.py:139-209); (b) the score
    and the IMQ phi of ALL particles against the oracle fed the device's costs (the stage-wise policy of the module docstring), compared
    at the 64 sampled rows and over the whole set.  IMQ itself has no reference (BASELINE names it, the upstream tree has none): the
    oracle's closed form is what it is held to."""
    from dust_amd import Context
    from oracle import Oracle

    N, S, M, H, ell = 2048, 128, 8, 30, 0.9
    rng = np.random.default_rng(2048)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    theta = (mu + 0.5 * rng.standard_normal((N, H, 1))).astype(np.float32)
    eps16 = rng.standard_normal((S, N, H, 1)).astype(np.float16)
    state = np.array([3.0, 0.0], np.float32)
    params = rng.uniform(0.6, 1.3, (M, 2)).astype(np.float32)
    c = Context(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=("length", "mass"), kernel="IMQ", imq_ell=ell, lr=2.0, alpha=1.0,
                sigma_a=2.0, sigma_p=2.0)
    c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
    costs = c.likelihood_sample(state, eps16, params)  # [S][N]: the full launch over binary16 noise
    phi, gl, gp = c.svmpc_phi()                        # score + IMQ phi from the stored costs / actions
    c.close()
    idx = np.sort(rng.choice(N, 64, replace=False))
    sg = np.full(1, 2.0, np.float32)
    eps = eps16.astype(np.float32)
    o64 = Oracle(model="pendulum", N=64, S=S, M=M, H=H, uncertain_params=("length", "mass"))
    ref = o64.rollout_cost(state, o64.sample_actions(theta[idx], np.ascontiguousarray(eps[:, idx]), sg), params)
    assert elemerr(costs[:, idx], ref) < TOL, elemerr(costs[:, idx], ref)
    o = Oracle(model="pendulum", N=N, S=S, M=M, H=H, uncertain_params=("length", "mass"))
    actions = o.sample_actions(theta, eps, sg)
    mix = np.full(N, 1.0 / N, np.float32)
    gl_r, gp_r, sc_r = o.score(theta, mu, mix, sg, costs, actions, 1.0, sg)
    assert elemerr(gl, gl_r) < TOL and elemerr(gp, gp_r) < TOL, (elemerr(gl, gl_r), elemerr(gp, gp_r))
    phi_r = o.phi_imq(theta, gl + gp, ell)
    assert elemerr(phi[idx], phi_r[idx]) < TOL and elemerr(phi, phi_r) < TOL, (elemerr(phi[idx], phi_r[idx]), elemerr(phi, phi_r))


@pytest.mark.parametrize("model,N,H,spread,offset", [("particle", 2048, 40, 1.0, 0.0), ("particle", 2100, 40, 0.25, 6.0),
                                                     ("pendulum", 4096, 30, 0.5, -3.0), ("pendulum", 2200, 17, 2.0, 0.0),
                                                     ("pendulum", 2048, 12, 0.05, 1.0), ("particle", 16384, 40, 1.0, 2.0)])
def test_large_aliased_logp_mfma_vs_oracle(model, N, H, spread, offset, monkeypatch):
    """SVMPC.forward's log p(theta) (svmpc.py:128-140) for N >= 2048 with the prior means aliasing theta runs the product-form
    distance on the matrix cores (pairwise_logp_mfma.hpp; rows centred on particle 0, so a common `offset` of the cloud costs no
    accuracy).  Against the oracle (exact differences, double accumulation) and against the exact-difference device pass
    (DUST_LOGP_MFMA=0): log p element-wise at 1e-5, particle weights at 1e-5 absolute.  Ragged N, D = 80 / 30 / 17 / 12,
    anisotropic sigma_p, non-uniform mixture weights, a nearly collapsed set (spread 0.05: every key matters), the cfg4 shape."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")  # (N D <= 65536 runs the small-set launches by default: this test is about the large-set kernels)
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(5 * N + H)
    S = 8
    theta = (offset + spread * rng.standard_normal((N, H, da))).astype(np.float32)
    mixw = rng.random(N).astype(np.float32) + 0.05
    mixw /= mixw.sum()
    grid = grid_4x4_map() if model == "particle" else None
    sg = np.full(da, 1.5, np.float32)
    sp = np.array([1.5, 0.8], np.float32)[:da]
    o = Oracle(model=model, N=N, S=S, M=1, H=H, grid=grid)
    alpha = 1.0 if model == "pendulum" else 1e-4
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    got = {}
    for mfma in ("1", "0"):
        os.environ["DUST_LOGP_MFMA"] = mfma
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel="K1", lr=0.0, alpha=alpha, sigma_a=sg, sigma_p=sp, grid=grid, weighted_prior=True,
                        seed=11)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)  # the means alias theta from here on
            c.svmpc_optimize(state, 1)  # lr = 0: the particles stay, the rollouts leave the costs the likelihood half needs
            assert np.array_equal(c.get_theta(), theta)
            pw = c.svmpc_get_weights()
            ll, lp = c.get_log_weights()
            got[mfma] = (lp, pw, c.get_costs())
            c.close()
        finally:
            os.environ.pop("DUST_LOGP_MFMA", None)
    lp, pw, costs = got["1"]
    lp0, pw0, costs0 = got["0"]
    assert np.array_equal(costs, costs0)
    fw = o.forward(costs, theta, theta, mixw, sp, alpha, weighted_prior=True)
    assert elemerr(lp0, fw["log_p"]) < 1e-5
    assert elemerr(lp, fw["log_p"]) < 1e-5, elemerr(lp, fw["log_p"])
    assert np.abs(lp - fw["log_p"]).max() < 2e-4, np.abs(lp - fw["log_p"]).max()   # absolute, on values of O(10-300)
    assert np.abs(pw - pw0).max() < 1e-6 + 3e-4 * pw0.max()  # same costs on both sides: only log p differs
    assert np.abs(pw - fw["p_weights"]).max() < 2e-3
    assert abs(float(pw.sum()) - 1.0) < 5e-4  # (log-weights of O(1e3): one fp32 ulp there is 1e-4 relative on a weight)


@pytest.mark.parametrize("model,N,H,kind", [("particle", 2048, 40, "spread"), ("particle", 2100, 40, "mixed"), ("pendulum", 4096, 30, "mixed"),
                                            ("pendulum", 2304, 30, "clustered"), ("pendulum", 2200, 17, "spread"),
                                            ("particle", 16384, 40, "mixed")])
def test_fused_pairwise_zero_blocks_bitwise(model, N, H, kind, monkeypatch):
    """K1's kernel values underflow to exactly 0 in fp32 beyond d^2 ~ 84; pairwise_fused.hpp flags the (query row, 64-key chunk)
    blocks that are all zero and skips them in pass B, in the Gram store and in the Gram x score GEMM.  The result must be
    BIT-IDENTICAL to the dense evaluation (DUST_DENSE=1): phi / grad_pri of the stage-wise call and the particles after two whole
    ticks, for a spread set (only the diagonal survives), a clustered one (everything survives) and a mixed one (a cluster of
    near-duplicates inside a spread set: blocks with a few non-zero rows - the rows of such a block that were NOT stored must read
    as zeros, not as stale Gram values of an earlier pass).  (pairwise_far.hpp at its exact-zero threshold: the default one trades
    bit-identity for terms below 2^-43 - test_fused_pairwise_far_units.)"""
    monkeypatch.setenv("DUST_FAR_T", "224")
    monkeypatch.setenv("DUST_PAIR_BIG", "1")  # (N D <= 65536 runs the small-set launches by default: this test is about the large-set kernels)
    from dust_amd import Context
    from oracle import grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(7 * N + H)
    S = 8
    sig = 2.0 if model == "pendulum" else 5.0
    theta = (sig * rng.standard_normal((N, H, da))).astype(np.float32)
    if kind == "clustered":
        theta = (0.05 * rng.standard_normal((N, H, da))).astype(np.float32)
    elif kind == "mixed":  # every 7th particle sits next to particle 3; a run of neighbours in index space as well
        theta[::7] = theta[3] + (0.3 * rng.standard_normal((len(theta[::7]), H, da))).astype(np.float32)
        theta[100:180] = theta[100] + (0.2 * rng.standard_normal((80, H, da))).astype(np.float32)
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, da))).astype(np.float32)
    mixw = rng.random(N).astype(np.float32) + 0.05
    grid = grid_4x4_map() if model == "particle" else None
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    got = {}
    for dense in ("", "1"):
        if dense:
            os.environ["DUST_DENSE"] = dense
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel="K1", lr=0.5, alpha=1.0 if da == 1 else 1e-4, sigma_a=sig, sigma_p=sig,
                        grid=grid, weighted_prior=True, seed=3)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)
            phi_d, _, gp_d = c.svmpc_phi(costs, actions)
            # the same call again after a pass over OTHER particles has left its Gram rows in the buffer
            c.set_theta(theta[::-1].copy())
            c.svmpc_phi(costs, actions)
            c.set_theta(theta)
            phi_2, _, gp_2 = c.svmpc_phi(costs, actions)
            assert np.array_equal(phi_2, phi_d) and np.array_equal(gp_2, gp_d)
            for _ in range(2):
                a_seq, pw = c.svmpc_tick(state, 1)
            got[dense] = (phi_d, gp_d, c.get_theta(), pw)
            c.close()
        finally:
            os.environ.pop("DUST_DENSE", None)
    for x, y in zip(got[""], got["1"]):
        assert np.array_equal(x, y)
    nzfrac = float((np.abs(got[""][0]) > 0).mean())
    assert nzfrac > 0.5  # (phi itself is dense: K_ii = 1 carries the score)


def _far_logp(ctx):
    import ctypes as C

    from dust_amd import _lib as L

    lib = L.load()
    lib.dust_debug_far_logp.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    lib.dust_debug_far_logp.restype = C.c_int
    out = (C.c_longlong * 2)()
    assert lib.dust_debug_far_logp(ctx._h, out) == 0
    return int(out[0]), int(out[1])


def _pack_lists(ctx, tile):
    """Run lists of one query tile of the context's last large-set pass 1 (dust_debug_pack_lists: a test hook, not part of the C ABI)."""
    import ctypes as C

    from dust_amd import _lib as L

    lib = L.load()
    IP, UP = C.POINTER(C.c_int), C.POINTER(C.c_uint)
    lib.dust_debug_pack_lists.argtypes = [C.c_void_p, C.c_int, IP, IP, IP, IP, UP, IP]
    lib.dust_debug_pack_lists.restype = C.c_int
    nu, js = C.c_int(), C.c_int()
    N = ctx.N
    uoff, kidx = np.zeros(N // 64 + 2, np.int32), np.zeros(N + 64, np.int32)
    uq, soff = np.zeros((N // 64 + 1) * 4, np.uint32), np.zeros(64, np.int32)
    assert lib.dust_debug_pack_lists(ctx._h, tile, C.byref(nu), C.byref(js), uoff.ctypes.data_as(IP), kidx.ctypes.data_as(IP), uq.ctypes.data_as(UP),
                                     soff.ctypes.data_as(IP)) == 0
    U, JS = nu.value, js.value
    return U, uoff[:U + 1].copy(), kidx[:uoff[U]].copy(), uq[:4 * U].reshape(U, 4).copy(), soff[:JS + 1].copy()


@pytest.mark.parametrize("N,shard", [(2100, None), (4096, None), (4096, (1024, 2048))])
def test_tile_order_is_a_permutation_and_changes_nothing_but_rounding(N, shard, monkeypatch):
    """pairwise_packed.hpp query_order_kernel: from the second tick on the large-set passes walk the queries in the order of their
    leader (the smallest close key index, noted by pass 1; one workgroup's stable radix sort).  The order must be a PERMUTATION of the
    rank's rows (ragged row counts, a shard in the middle of the set) - a row listed twice leaves another out of both passes - and it only
    groups work: five ticks with the order and five with index order (DUST_PACK_ORDER=0) agree to the tolerance of a regrouped sum."""
    import ctypes as C

    monkeypatch.setenv("DUST_PAIR_BIG", "1")
    from dust_amd import Context
    from dust_amd import _lib as L
    from oracle import grid_4x4_map

    H, S = 40, 8
    rng = np.random.default_rng(N)
    theta = (3.0 * rng.standard_normal((N, H, 2))).astype(np.float32)
    theta[::5] = theta[2] + (0.4 * rng.standard_normal((len(theta[::5]), H, 2))).astype(np.float32)   # scattered near-duplicates
    theta[300:700] = theta[300] + (0.5 * rng.standard_normal((400, H, 2))).astype(np.float32)
    state = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
    kw = dict(model="particle", N=N, S=S, M=1, H=H, kernel="K1", lr=0.5, alpha=1e-4, sigma_a=2.0, sigma_p=1.0, grid=grid_4x4_map(), seed=3)
    if shard:
        kw.update(shard_offset=shard[0], shard_size=shard[1])
    out = {}
    for order in ("1", "0"):
        monkeypatch.setenv("DUST_PACK_ORDER", order)
        c = Context(**kw)
        c.set_theta(theta); c.set_prior(theta); c.set_a_mat(theta)
        if shard:  # a rank alone (the other rows stay as they are): local score + Stein step + forward, the rank's kernels
            lib = L.load()
            for _ in range(5):
                L.check(lib.dust_svmpc_local_score(c._h, state.ctypes.data_as(L.FP), None, None, 0))
                L.check(lib.dust_svmpc_apply_phi(c._h))
                lw, nb = L.VP(), C.c_size_t(0)
                L.check(lib.dust_svmpc_forward_local(c._h, C.byref(lw), C.byref(nb)))
                L.check(lib.dust_svmpc_forward_finish(c._h, None, None))
        else:
            for _ in range(5):
                c.svmpc_tick(state, 1)
        if order == "1":
            lib = L.load()
            lib.dust_debug_tile_order.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
            lib.dust_debug_tile_order.restype = C.c_int
            n_loc = shard[1] if shard else N
            perm = np.zeros(n_loc, np.int32)
            assert lib.dust_debug_tile_order(c._h, perm.ctypes.data_as(C.POINTER(C.c_int))) == 0
            assert np.array_equal(np.sort(perm), np.arange(n_loc)), "the tile order is not a permutation of the rank's rows"
            assert not np.array_equal(perm, np.arange(n_loc)), "leaders were noted: the order should differ from index order on this set"
        out[order] = c.get_theta()
        c.close()
    rows = slice(shard[0], shard[0] + shard[1]) if shard else slice(None)
    assert elemerr(out["1"][rows], out["0"][rows]) < 2e-5, elemerr(out["1"][rows], out["0"][rows])


@pytest.mark.parametrize("mode", ["merged", "plain"])
def test_run_lists_cover_every_near_pair(mode, monkeypatch):
    """pairwise_packed.hpp far_pack_kernel: a query tile's run list must hold EVERY key with a near query in the tile (the pre-pass'
    bound is one-sided: a pair whose exact scaled distance is under the threshold is never called far), ascending and without
    repeats; a unit's query mask must hold every query with a near key in the unit; the slices partition the units.  Merged lists
    (default threshold) pack 64 listed keys per unit; plain lists (exact-zero threshold) are whole live chunks."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")
    if mode == "plain":
        monkeypatch.setenv("DUST_FAR_T", "224")
    from dust_amd import Context
    from oracle import grid_4x4_map

    N, H, S = 2100, 40, 8
    rng = np.random.default_rng(17)
    theta = (3.0 * rng.standard_normal((N, H, 2))).astype(np.float32)
    theta[:512:7] = theta[3] + (0.3 * rng.standard_normal((len(theta[:512:7]), H, 2))).astype(np.float32)
    theta[1000:1400] = theta[1000] + (0.6 * rng.standard_normal((400, H, 2))).astype(np.float32)
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, 2))).astype(np.float32)
    c = Context(model="particle", N=N, S=S, M=1, H=H, kernel="K1", lr=0.5, alpha=1e-4, sigma_a=2.0, sigma_p=1.0, grid=grid_4x4_map(), seed=3)
    c.set_theta(theta)
    c.set_prior(theta)
    c.set_a_mat(theta)
    c.svmpc_update_prior(np.ones(N, np.float32))  # uniform weights: a pair is near iff its scaled distance is under the threshold
    c.svmpc_phi(costs, actions)
    X = theta.reshape(N, -1).astype(np.float64)  # g = min(1 / ell^2, 1 / sigma_p^2) = 1
    n2 = (X * X).sum(1)
    T = 224.0 if mode == "plain" else 60.0
    TQ = 96
    total_units = 0
    for tile in (0, 1, 5, 10, 14, 21):
        q = np.arange(tile * TQ, min(N, (tile + 1) * TQ))
        G = n2[q, None] + n2[None, :] - 2.0 * X[q] @ X.T
        near = G <= T
        U, uoff, kidx, uq, soff = _pack_lists(c, tile)
        total_units += U
        assert (np.diff(kidx) > 0).all() and kidx.min() >= 0 and kidx.max() < N
        must = np.flatnonzero(near.any(axis=0))
        assert np.isin(must, kidx).all(), (tile, np.setdiff1d(must, kidx)[:8])
        assert uoff[0] == 0 and (np.diff(uoff) > 0).all() and (np.diff(uoff) <= 64).all()
        assert soff[0] == 0 and soff[-1] == U and (np.diff(soff) >= 0).all()
        if mode == "merged":
            assert (np.diff(uoff)[:-1] == 64).all()  # packed: only the last unit may be ragged
            assert len(kidx) < N // 2  # (a mixed set: most keys have no near query in a tile of 96)
        for u in range(U):
            keys = kidx[uoff[u]:uoff[u + 1]]
            if mode == "plain":  # a whole chunk at its own lane positions
                assert keys[0] % 64 == 0 and (np.diff(keys) == 1).all() and len(keys) == min(64, N - keys[0])
            bits = np.array([(int(uq[u, i // 32]) >> (i % 32)) & 1 for i in range(len(q))], bool)
            need = near[:, keys].any(axis=1)
            assert (bits | ~need).all(), (tile, u)
    assert total_units > 0
    c.close()


@pytest.mark.parametrize("model,N,H,kind,weights", [
    ("particle", 2048, 40, "spread", "flat"), ("particle", 2100, 40, "mixed", "steep"), ("particle", 2048, 40, "mixed", "zeros"),
    ("pendulum", 2304, 30, "mixed", "steep"), ("particle", 16384, 40, "mixed", "steep"), ("particle", 2048, 40, "clustered", "flat")])
def test_logp_far_blocks(model, N, H, kind, weights, monkeypatch):
    """The log-p pass (pairwise_logp_mfma.hpp) leaves out the (64-query group, key chunk) blocks pairwise_far.hpp proves negligible
    (every term below 2^-43 of the query's largest): log p against the run that visits everything (DUST_FAR=0) to 1e-6 relative /
    2e-5 absolute on values of O(10 - 1000) (the oracle comparison of the pass itself: test_large_aliased_logp_mfma_vs_oracle); weights spanning
    e^-60 and exact zeros."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")
    from dust_amd import Context
    from oracle import Oracle, grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(13 * N + H)
    S = 8
    theta = ((3.0 if da == 2 else 8.0) * rng.standard_normal((N, H, da))).astype(np.float32)
    if kind == "clustered":
        theta = (0.05 * rng.standard_normal((N, H, da))).astype(np.float32)
    elif kind == "mixed":
        theta[:512:7] = theta[3] + (0.3 * rng.standard_normal((len(theta[:512:7]), H, da))).astype(np.float32)
        theta[100:180] = theta[100] + (0.2 * rng.standard_normal((80, H, da))).astype(np.float32)
        theta[1000:1400] = theta[1000] + (1.0 * rng.standard_normal((400, H, da))).astype(np.float32)
    mixw = rng.random(N).astype(np.float32) + 0.05
    if weights == "steep":
        mixw = np.exp(-60.0 * rng.random(N)).astype(np.float32)
    elif weights == "zeros":
        mixw[::5] = 0.0
        mixw[101] = 0.0
    mixw /= mixw.sum()
    grid = grid_4x4_map() if model == "particle" else None
    sp = np.array([1.0, 0.8], np.float32)[:da] if da == 2 else 0.5
    alpha = 1.0 if model == "pendulum" else 1e-4
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    got, far = {}, {}
    for mode in ("far", "nofar"):
        if mode == "nofar":
            os.environ["DUST_FAR"] = "0"
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel="K1", lr=0.0, alpha=alpha, sigma_a=2.0, sigma_p=sp, grid=grid, weighted_prior=True,
                        seed=11)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)
            c.svmpc_optimize(state, 1)
            pw = c.svmpc_get_weights()
            far[mode] = _far_logp(c)
            ll, lp = c.get_log_weights()
            got[mode] = (lp, pw, c.get_costs())
            c.close()
        finally:
            os.environ.pop("DUST_FAR", None)
    lp, pw, costs = got["far"]
    lp0, pw0, costs0 = got["nofar"]
    assert np.array_equal(costs, costs0)
    fin = np.isfinite(lp0)
    assert np.array_equal(fin, np.isfinite(lp)) and fin.mean() > 0.5
    assert np.abs(lp[fin] - lp0[fin]).max() < 2e-5 + 1e-6 * np.abs(lp0[fin]).max(), np.abs(lp[fin] - lp0[fin]).max()
    assert np.abs(pw - pw0).max() < 1e-6 + 1e-5 * pw0.max()
    f, u = far["far"]
    assert far["nofar"] == (0, 0) and u > 0
    if kind == "spread":
        assert f > 0.7 * u, (f, u)
    elif kind == "mixed":
        assert 0.05 * u < f < u, (f, u)
    else:
        assert f == 0, (f, u)


def _far_units(ctx):
    """(far, all) units of the context's last fused pairwise pass (dust_debug_far_units: not part of the C ABI)."""
    import ctypes as C

    from dust_amd import _lib as L

    lib = L.load()
    lib.dust_debug_far_units.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    lib.dust_debug_far_units.restype = C.c_int
    out = (C.c_longlong * 2)()
    assert lib.dust_debug_far_units(ctx._h, out) == 0
    return int(out[0]), int(out[1])


@pytest.mark.parametrize("model,N,H,kind,weights", [
    ("particle", 2048, 40, "spread", "flat"), ("particle", 2100, 40, "mixed", "steep"), ("particle", 2048, 40, "mixed", "zeros"),
    ("pendulum", 2304, 30, "mixed", "steep"), ("pendulum", 2200, 17, "spread", "flat"), ("particle", 2048, 16, "mixed", "flat"),
    ("particle", 16384, 40, "mixed", "steep"), ("particle", 2048, 40, "clustered", "flat"), ("particle", 2048, 40, "huge", "flat")])
def test_fused_pairwise_far_units(model, N, H, kind, weights, monkeypatch):
    """pairwise_far.hpp: (query tile, key chunk) units whose Stein kernel values AND prior softmax terms are all negligible are found by
    a binary16 GEMM bound and never visited by the exact-difference pass.  At the exact-zero threshold (DUST_FAR_T=224) skipping must
    be BIT-IDENTICAL to visiting them (DUST_FAR=0) and to the dense evaluation (DUST_DENSE=1); at the default threshold (terms
    below 2^-43 of their sum's leading term) within 1e-6 of the largest element: phi / grad_pri of the stage-wise call, the
    particles and weights after two whole ticks - for a spread set (everything off the diagonal is far), a mixed one (clusters of
    near-duplicates inside a spread set, in index runs and scattered), a clustered one (nothing is far), mixture weights spanning
    e^-60 and exact zeros (the prior's far test carries log w_j - m0_i), and coordinates beyond binary16 (never far)."""
    monkeypatch.setenv("DUST_PAIR_BIG", "1")
    from dust_amd import Context
    from oracle import grid_4x4_map

    da = 1 if model == "pendulum" else 2
    rng = np.random.default_rng(11 * N + H)
    S = 8
    sig_p = 1.0 if da == 2 else 0.5
    theta = ((3.0 if da == 2 else 8.0) * rng.standard_normal((N, H, da))).astype(np.float32)
    if kind == "clustered":
        theta = (0.05 * rng.standard_normal((N, H, da))).astype(np.float32)
    elif kind == "mixed":
        theta[:512:7] = theta[3] + (0.3 * rng.standard_normal((len(theta[:512:7]), H, da))).astype(np.float32)
        theta[100:180] = theta[100] + (0.2 * rng.standard_normal((80, H, da))).astype(np.float32)
        theta[1000:1400] = theta[1000] + (1.0 * rng.standard_normal((400, H, da))).astype(np.float32)  # around the thresholds
    elif kind == "huge":
        theta[5] *= 1.0e5
        theta[777, 3] = 3.0e7
    mixw = rng.random(N).astype(np.float32) + 0.05
    if weights == "steep":
        mixw = np.exp(-60.0 * rng.random(N)).astype(np.float32)
    elif weights == "zeros":
        mixw[::5] = 0.0
        mixw[101] = 0.0
    costs = (30.0 * rng.random((S, N))).astype(np.float32)
    actions = (theta[None] + rng.standard_normal((S, N, H, da))).astype(np.float32)
    grid = grid_4x4_map() if model == "particle" else None
    state = np.array([3.0, 0.0] if da == 1 else [-9.0, -9.0, 0.0, 0.0], np.float32)
    got, far = {}, {}
    for mode in ("far", "far_plain", "exact", "nofar", "dense"):
        if mode == "far_plain":  # the default threshold on PLAIN run lists (pairwise_packed.hpp): what is left out, without the regrouping
            os.environ["DUST_PACK_MERGE"] = "0"
        if mode == "exact":
            os.environ["DUST_FAR_T"] = "224"
        if mode == "nofar":
            os.environ["DUST_FAR"] = "0"
        if mode == "dense":
            os.environ["DUST_DENSE"] = "1"
        try:
            c = Context(model=model, N=N, S=S, M=1, H=H, kernel="K1", lr=0.5, alpha=1.0 if da == 1 else 1e-4, sigma_a=2.0, sigma_p=sig_p,
                        grid=grid, weighted_prior=True, seed=3)
            c.set_theta(theta)
            c.set_prior(theta)
            c.set_a_mat(theta)
            c.svmpc_update_prior(mixw)
            phi_d, _, gp_d = c.svmpc_phi(costs, actions)
            far[mode] = _far_units(c)
            c.set_theta(theta[::-1].copy())  # a pass over OTHER particles leaves its Gram rows and flags in the buffers
            c.svmpc_phi(costs, actions)
            c.set_theta(theta)
            phi_2, _, gp_2 = c.svmpc_phi(costs, actions)
            assert np.array_equal(phi_2, phi_d, equal_nan=True) and np.array_equal(gp_2, gp_d, equal_nan=True)
            pw = None
            if kind != "huge":
                for _ in range(2):
                    a_seq, pw = c.svmpc_tick(state, 1)
            got[mode] = (phi_d, gp_d, c.get_theta(), pw)
            c.close()
        finally:
            os.environ.pop("DUST_FAR", None)
            os.environ.pop("DUST_FAR_T", None)
            os.environ.pop("DUST_DENSE", None)
            os.environ.pop("DUST_PACK_MERGE", None)
    for other in ("nofar", "dense"):
        for x, y in zip(got["exact"], got[other]):
            assert (x is None and y is None) or np.array_equal(x, y, equal_nan=True), other
    if kind != "huge":
        # (grad_pri of a spread set consists of nothing BUT negligible terms - e^-60 and below: held to an absolute floor, as the
        #  score it is added to is O(1); otherwise to its own RMS, as every sum with cancellation is - tests/helpers.py: the run lists
        #  of the default mode regroup the sums over a tile's keys, pairwise_packed.hpp)
        # Two effects, held apart.  What the default threshold LEAVES OUT (terms below 2^-43 of their sum's leading term): the run that
        # leaves them out on plain lists against the run that visits everything, 1e-6 on all four.  The merged lists also REGROUP the
        # sums over a tile's keys (chunk-wise online softmax, 4-key MFMA steps, slice partials): rounding of a different summation
        # order - 2e-6 on the stage-wise phi / grad_pri, and on what two whole ticks make of it 1e-5 (particles) / 2e-3 (the weights:
        # softmax(-alpha cost + log p) amplifies a cost ulp, the whole-tick bound of every other test)
        for k, (x, y) in enumerate(zip(got["far_plain"], got["nofar"])):
            floor = max(1e-3, float(np.sqrt(np.mean(np.float64(y) ** 2)))) if k == 1 else None
            assert elemerr(x, y, floor=floor) < 1e-6, ("plain lists", k, elemerr(x, y))
        for k, (x, y) in enumerate(zip(got["far"], got["nofar"])):
            floor = max(1e-3, float(np.sqrt(np.mean(np.float64(y) ** 2)))) if k == 1 else None
            assert elemerr(x, y, floor=floor) < (2e-6, 2e-6, 1e-5, 2e-3)[k], ("merged lists", k, elemerr(x, y))
    f, u = far["far"]
    fe, _ = far["exact"]
    assert far["nofar"] == (0, 0) and far["dense"] == (0, 0) and u > 0 and fe <= f and far["far_plain"] == far["far"]
    if kind in ("spread", "huge"):
        assert f > 0.7 * u, (f, u, fe)
    elif kind == "mixed":
        assert 0.05 * u < f < u, (f, u, fe)
    else:
        assert f == 0, (f, u)


@pytest.mark.parametrize("name", ["mpf_bwvec", "mpf_bwiqr"])
def test_mpf_initial_prior_from_bw_silverman(golden, name):
    """MPF(bw=None) with P = 2 (mpf.py:29-38): the first prior is diag(bw_silverman(columns)^2) - per-dimension bandwidths on the
    device (`dust_mpf_set_prior_bw`), through the raw context and through the mirror class `dust_amd.inference.mpf.MPF(bw=None)`;
    golden = the reference's own MPF (tests/golden/make_golden_r3.py)."""
    import torch

    from dust_amd import MpfContext
    from dust_amd.inference.likelihoods import GaussianLikelihood
    from dust_amd.inference.mpf import MPF
    from dust_amd.inference.svgd import bw_silverman
    from dust_amd.models import PendulumModel

    g = golden(name)
    P, bw, lr, n = int(g["P"]), float(g["bw_opt"]), float(g["lr"]), int(g["n_steps"])
    bwv = np.broadcast_to(np.asarray(bw_silverman(g["x0"], 1.0), np.float32).reshape(-1), (P,)).astype(np.float32)
    tol_x = TOL * (1.0 + lr / float(bwv.min()) ** 2) ** n  # (stiff attraction of the prior: tests/test_oracle_golden.py, same test)
    up = ("length", "mass")
    m = MpfContext(g["x0"], g["obs0"], model="pendulum", uncertain_params=up, log_space=False, obs_std=float(g["obs_std"]), lr=lr,
                   init_bw=float(bwv[0]))
    m.set_prior_bw(bwv)
    assert np.array_equal(m.get_prior_bw(), bwv)
    assert relerr(m.prior_log_prob(g["probe"]), g["probe_log_prob0"]) < TOL
    smp = m.prior_sample(40000, seed=5)
    spread = smp.std(0) ** 2 - g["x0"].std(0) ** 2  # mixture variance = variance of the means + bw_p^2
    assert np.all(np.abs(spread - bwv ** 2) < 0.15 * bwv ** 2 + 0.05 * g["x0"].std(0) ** 2)
    m.condition(g["action"], g["obs1"])
    assert elemerr(m.phi(bw), g["phi0"]) < TOL
    m.close()
    # the mirror class: bw=None evaluates bw_silverman on the host and hands the vector over
    model = PendulumModel(uncertain_params=up)
    lik = GaussianLikelihood(initial_obs=torch.tensor(g["obs0"]), obs_std=float(g["obs_std"]), model=model, log_space=False)
    mp = MPF(init_particles=torch.tensor(g["x0"]), likelihood=lik, optimizer_class=torch.optim.SGD, lr=lr, bw=None, bw_scale=1.0)
    assert relerr(mp.prior.log_prob(torch.tensor(g["probe"])).numpy(), g["probe_log_prob0"]) < TOL
    grads, _ = mp.optimize(torch.tensor(g["action"]), torch.tensor(g["obs1"]), bw=bw, n_steps=n)
    assert relerr(mp.x.numpy(), g["x_final"]) < tol_x and relerr(grads.numpy(), g["grad_norms"]) < TOL
    assert relerr(mp.prior.log_prob(torch.tensor(g["probe"])).numpy(), g["probe_log_prob1"]) < TOL  # isotropic again (mpf.py:85)
    grads2, _ = mp.optimize(torch.tensor(g["action2"]), torch.tensor(g["obs2"]), bw=bw, n_steps=n)
    assert relerr(mp.x.numpy(), g["x_final2"]) < 2 * tol_x and relerr(grads2.numpy(), g["grad_norms2"]) < 2e-4


@pytest.mark.parametrize("form", ["poll", "counter"])
@pytest.mark.parametrize("name", ["mpf_pend", "mpf_part_log", "mpf_pend_adam", "mpf_part_log_adam"])
def test_mpf_multi_workgroup_kernel_vs_reference(golden, name, form, monkeypatch):
    """The multi-workgroup form of MPF.optimize (mpf.hpp mpf_optimize_grid_kernel: one wave per particle, two grid-wide hand-offs per
    step; the default from 96 particles on) against the reference's own filter updates - DUST_MPF_GRID=1 sends these small golden
    cases through it."""
    from dust_amd import MpfContext
    from oracle import grid_4x4_map

    monkeypatch.setenv("DUST_MPF_GRID", "1")
    monkeypatch.setenv("DUST_MPF_POLL", "1" if form == "poll" else "0")
    g = golden(name)
    kind = str(g["model_kind"])
    up = ("length", "mass") if kind == "pendulum" else ("mass",)
    bw, ls, n = float(g["bw"]), bool(int(g["log_space"])), int(g["n_steps"])
    m = MpfContext(g["x0"], g["obs0"], model=kind, uncertain_params=up, log_space=ls, obs_std=float(g["obs_std"]), lr=float(g["lr"]),
                   init_bw=bw, grid=grid_4x4_map() if kind == "particle" else None, mass=2.0 if kind == "particle" else 1.0,
                   optimizer="Adam" if name.endswith("adam") else "SGD")
    gn = m.optimize(g["action"], g["obs1"], bw, n)
    assert elemerr(m.get_particles(), g["x_final"]) < TOL
    assert relerr(gn, g["grad_norms"]) < 2e-4
    gn2 = m.optimize(g["action2"], g["obs2"], bw, n)
    assert elemerr(m.get_particles(), g["x_final2"]) < TOL
    assert relerr(gn2, g["grad_norms2"]) < 2e-4
    assert m.stats() == {"grid": 2, "fallback": 0}


@pytest.mark.parametrize("kind,Mp,opt,hook,form", [
    ("pendulum", 256, "SGD", None, "poll"), ("pendulum", 130, "Adam", None, "poll"), ("particle", 128, "SGD", None, "poll"),
    ("pendulum", 500, "Adam", None, "poll"), ("pendulum", 41, "SGD", None, "poll"), ("pendulum", 1024, "SGD", None, "poll"),
    ("pendulum", 1021, "Adam", None, "poll"), ("particle", 600, "Adam", None, "poll"), ("pendulum", 700, "SGD", "2", "poll"),
    ("pendulum", 256, "SGD", None, "counter"), ("pendulum", 1024, "SGD", None, "counter"), ("pendulum", 1021, "Adam", None, "counter"),
    ("pendulum", 256, "Adam", "1", "poll"), ("pendulum", 200, "SGD", "2", "poll"), ("pendulum", 200, "Adam", "2", "counter")])
def test_mpf_multi_workgroup_kernel_vs_single(kind, Mp, opt, hook, form, monkeypatch):
    """From 96 particles on MPF.optimize runs spread over the chip (data-polled exchange by default, arrival counters with DUST_MPF_POLL=0;
    `form` pins one of them); the single-workgroup kernel (DUST_MPF_GRID=0) does the same
    arithmetic per particle and sums in another fixed order.  Two filter updates each; ragged particle counts; both optimisers.
    hook 1 / 2: the grid form aborts at its start barrier / one of its waits "gives up" before the last hand-off - nothing is
    committed, the single-workgroup kernel runs the call, the caller sees DUST_OK and the same numbers."""
    from dust_amd import MpfContext
    from oracle import grid_4x4_map

    rng = np.random.default_rng(Mp)
    pend = kind == "pendulum"
    up = ("length", "mass") if pend else ("mass",)
    x0 = (1.0 + 0.2 * rng.standard_normal((Mp, len(up)))).astype(np.float32)
    obs = [np.array([3.0, 0.0], np.float32), np.array([3.02, 0.41], np.float32), np.array([3.08, 0.93], np.float32)] if pend else \
          [np.array([0.0, 0.0, 0.0, 0.0], np.float32), np.array([0.001, 0.002, 0.05, 0.1], np.float32), np.array([0.004, 0.01, 0.1, 0.2], np.float32)]
    acts = [np.array([1.0], np.float32), np.array([-0.5], np.float32)] if pend else [np.array([1.0, 2.0], np.float32), np.array([1.0, 2.0], np.float32)]
    out = []
    for grid in ("0", None):
        monkeypatch.delenv("DUST_MPF_GRID", raising=False)
        monkeypatch.delenv("DUST_MPF_GRID_TEST", raising=False)
        monkeypatch.setenv("DUST_MPF_GRID", grid or "1")  # (1: the grid form whatever the particle count; its default starts at 96)
        monkeypatch.setenv("DUST_MPF_POLL", "1" if form == "poll" else "0")
        if grid is None and hook:
            monkeypatch.setenv("DUST_MPF_GRID_TEST", hook)
        m = MpfContext(x0, obs[0], model=kind, uncertain_params=up, obs_std=0.1, lr=1e-3, init_bw=0.1, optimizer=opt,
                       grid=None if pend else grid_4x4_map(), mass=1.0)
        gn1 = m.optimize(acts[0], obs[1], 0.1, 20)
        x1 = m.get_particles()
        gn2 = m.optimize(acts[1], obs[2], 0.12, 7)
        out.append((x1, m.get_particles(), gn1, gn2, m.stats()))
        m.close()
    (a1, a2, ga1, ga2, sa), (b1, b2, gb1, gb2, sb) = out
    assert sa == {"grid": 0, "fallback": 0}
    # hook 2 ("a wait gave up") takes the context off the grid form for good: the second call does not try it again
    assert sb == ({"grid": 2, "fallback": 0} if hook is None else {"grid": 2, "fallback": 2} if hook == "1" else {"grid": 1, "fallback": 1}), sb
    if hook:
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and np.array_equal(ga1, gb1)
    else:
        assert np.isfinite(b2).all()
        assert elemerr(b1, a1) < 1e-5 and elemerr(b2, a2) < 1e-5, (elemerr(b1, a1), elemerr(b2, a2))
        assert relerr(gb1, ga1) < 1e-5 and relerr(gb2, ga2) < 1e-5


def _skid_ctx(g, N, S, M, H, up=(), **kw):
    from dust_amd import Context

    return Context(model="skid_steer", N=N, S=S, M=M, H=H, dt=float(g["dt"]), sigma_a=float(g["sigma_a"]), sigma_p=float(g["sigma_a"]),
                   uncertain_params=up or None, params_log_space=bool(int(g["params_log_space"])), goal=g["goal"], w_quad_state=g["w_state"],
                   w_quad_term=g["w_term"], w_quad_ctrl=g["w_ctrl"], **kw)


@pytest.mark.parametrize("name", ["skid_nominal", "skid_params", "skid_params_log"])
def test_skid_steer_family_vs_reference(golden, name):
    """the skid-steer rollout family on the device (csrc/skid.hpp + the regular kernel's weights stage) against the
    reference's own MultiDISCO.forward on SkidSteerRobot with the quadratic cost: costs, all states, omega, the a_mat update, a_mix."""
    g = golden(name)
    N, H, S, M = int(g["N"]), int(g["H"]), int(g["S"]), int(g["M"])
    up = ("x_icr", "wheel_radius") if "params" in g else ()
    temp = float(g["temperature"])
    c = _skid_ctx(g, N, S, M, H, up, temperature=temp, alpha=1.0 / temp)
    c.set_a_mat(g["a_mat0"])
    costs, states, _, omega = c.disco_forward(g["state"], g["ext_actions"], params=g["params"] if up else None, want_states=True)
    assert elemerr(costs, g["costs"]) < TOL
    assert np.abs(states - g["states"]).max() < 1e-5 * max(1.0, np.abs(g["states"]).max())
    assert relerr(omega, g["omega"]) < 2e-4
    assert relerr(c.get_a_mat(), g["a_mat1"]) < 1e-4
    assert relerr(c.get_a_mix(), g["a_mix"]) < 2e-4
    c.close()


def test_skid_steer_svmpc_ticks_vs_oracle(golden):
    """Whole SVGD-MPC ticks on the skid-steer family (launch-per-iteration path): caller-supplied noise and sampled parameters
    against the oracle's composition (skid rollouts -> score -> K1 phi -> SGD -> forward); the device Philox draws fetched as actions
    and replayed through the oracle; a clone continues identically."""
    from oracle import Oracle

    g = golden("skid_params")
    N, S, H, M, K = 12, 32, 10, 3, 2
    rng = np.random.default_rng(3)
    up = ("x_icr", "wheel_radius")
    sig, lr, alpha = 0.3, 0.05, 0.5
    mu = (0.2 * rng.standard_normal((N, H, 2))).astype(np.float32)
    theta = (mu + 0.1 * rng.standard_normal((N, H, 2))).astype(np.float32)
    state = g["state"]
    c = _skid_ctx(g, N, S, M, H, up, kernel="K1", lr=lr, alpha=alpha, seed=9)
    c.set_theta(theta); c.set_prior(mu); c.set_a_mat(theta)
    eps = rng.standard_normal((K, S, N, H, 2)).astype(np.float32)
    params = np.stack([rng.uniform([0.1, 0.05], [0.3, 0.08], (M, 2)) for _ in range(K)]).astype(np.float32)
    a_seq, pw = c.svmpc_tick(state, K, eps=eps, params=params)
    assert c.tick_stats()["tick2"] == 0
    o = Oracle(model="particle", N=N, S=S, M=1, H=H)  # (score / phi / forward do not touch the model)
    sg = np.full(2, sig, np.float32)
    kw = dict(uncertain_params=up, dt=float(g["dt"]), goal=g["goal"], w_state=g["w_state"], w_term=g["w_term"], w_ctrl=g["w_ctrl"])
    th, mix = theta.copy(), np.ones(N, np.float32)
    for k in range(K):
        actions = o.sample_actions(th, eps[k], sg)
        costs = Oracle.skid_rollout_cost(state, actions, params=params[k], **kw)
        if k == K - 1:
            assert elemerr(c.get_costs(), costs) < TOL
        _, _, sc = o.score(th, mu, mix, sg, costs, actions, alpha, sg)
        th = o.sgd(th, o.phi_k1(th, sc), lr)
    r = o.forward(costs, th, mu, mix, sg, alpha)
    assert elemerr(c.get_theta(), r["theta"]) < 1e-4
    assert np.abs(pw - r["p_weights"]).max() < 2e-3
    # device Philox noise: fetch the draws as actions, replay the rollouts through the oracle
    cc = c.clone()
    costs_dev, actions = c.likelihood_sample(state, None, params[0], want_actions=True)
    assert elemerr(costs_dev, Oracle.skid_rollout_cost(state, actions, params=params[0], **kw)) < TOL
    z = (actions - c.get_theta()[None]) / sig
    assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1.0) < 0.05
    a1, p1 = c.svmpc_tick(state, 2, params=params)
    cc.likelihood_sample(state, None, params[0])  # (the same stream position as c)
    a2, p2 = cc.svmpc_tick(state, 2, params=params)
    assert np.array_equal(a1, a2) and np.array_equal(p1, p2)
    c.close(); cc.close()


def test_skid_steer_sharded_equals_unsharded(golden):
    """The skid-steer family under particle sharding (2 and 4 sharded contexts in one process, all-gathers as slice copies): its
    rollout kernel takes (shard offset, local count) like every other kernel."""
    from dust_amd.parallel import DeviceShard, LocalComm, tick

    g = golden("skid_params")
    N, S, H, M, K, T = 64, 32, 10, 3, 2, 2
    rng = np.random.default_rng(5)
    up = ("x_icr", "wheel_radius")
    mu = (0.2 * rng.standard_normal((N, H, 2))).astype(np.float32)
    th = (mu + 0.1 * rng.standard_normal((N, H, 2))).astype(np.float32)
    state = g["state"]
    eps = rng.standard_normal((T, K, S, N, H, 2)).astype(np.float32)
    params = np.stack([[rng.uniform([0.1, 0.05], [0.3, 0.08], (M, 2)) for _ in range(K)] for _ in range(T)]).astype(np.float32)
    kw = dict(model="skid_steer", N=N, S=S, M=M, H=H, dt=float(g["dt"]), sigma_a=0.3, sigma_p=0.3, uncertain_params=up, goal=g["goal"],
              w_quad_state=g["w_state"], w_quad_term=g["w_term"], w_quad_ctrl=g["w_ctrl"], kernel="K1", lr=0.05, alpha=0.5, seed=11)
    ref = _skid_ctx(g, N, S, M, H, up, kernel="K1", lr=0.05, alpha=0.5, seed=11)
    ref.set_theta(th); ref.set_prior(mu); ref.set_a_mat(th)
    outs = [ref.svmpc_tick(state, K, eps[t], params[t]) for t in range(T)]
    rt = ref.get_theta()
    for world in (2, 4):
        shards = tuple(DeviceShard(dict(kw), r, world) for r in range(world))
        for sh in shards:
            sh.set_state(th, mu, th)
        for t in range(T):
            a_seq, pw = tick(shards, LocalComm(), state, K, eps[t], params[t], want_outputs=True, final_gather=(world == 2))
            assert np.array_equal(a_seq, outs[t][0]), (world, t)
            assert relerr(pw, outs[t][1]) < 1e-5
        for sh in shards:
            sh.sync()
            assert elemerr(sh.ctx.get_theta(), rt) < 2e-6, (world, sh.rank)
            sh.ctx.close()
    ref.close()


def test_device_noise_generator_statistics():
    """philox_normal8 (common.hpp) - the policy-noise generator of every device path since round 4 - builds each Box-Muller pair
    from two 16-bit uniforms: 65 536 radii up to 4.85 sigma.  What that does to the DISTRIBUTION, measured on 3.9 M draws of the product
    shape (fetched as actions around a_mat = 0 with sigma_a = 1): moments to 5 standard errors, the tail masses beyond 3 and 4 sigma to
    5 binomial standard deviations, the hard cut at sqrt(-2 ln 2^-17) = 4.855, and no correlation between neighbouring columns (the two
    halves of a pair, consecutive pairs) or between samples."""
    from math import erfc, sqrt

    from dust_amd import Context

    N, S, H = 1024, 128, 30
    c = Context(model="pendulum", N=N, S=S, M=1, H=H, sigma_a=1.0, sigma_p=1.0, seed=20260101)
    zs = []
    for _ in range(2):  # two launches: two positions of the counter stream
        c.set_a_mat(np.zeros((N, H, 1), np.float32))  # (forward's side effect moves a_mat: disco.py:387-392)
        _, _, act, _ = c.disco_forward(np.array([3.0, 0.0], np.float32), None, want_actions=True)
        zs.append(act.reshape(S, N, H).astype(np.float64))
    c.close()
    assert not np.array_equal(zs[0], zs[1])
    z = np.concatenate([a.reshape(-1) for a in zs])
    n = z.size
    assert abs(z.mean()) < 5 / sqrt(n)
    assert abs(z.var() - 1.0) < 5 * sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 5 * sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3.0) < 5 * sqrt(96.0 / n) + 2e-3  # (the tail cut takes ~1e-3 off the fourth moment)
    for t in (3.0, 4.0):
        p = erfc(t / sqrt(2.0))
        k = float((np.abs(z) > t).sum())
        assert abs(k - n * p) < 5 * sqrt(n * p) + 1, (t, k, n * p)
    assert np.abs(z).max() <= 4.8553 + 1e-3, np.abs(z).max()
    a = zs[0]
    for lag in (1, 2, 8):  # neighbouring columns of a row: the sin / cos halves of one pair, consecutive pairs, consecutive Philox blocks
        r = np.corrcoef(a[:, :, :-lag].reshape(-1), a[:, :, lag:].reshape(-1))[0, 1]
        assert abs(r) < 5 / sqrt(a[:, :, lag:].size), (lag, r)
    r = np.corrcoef(a[:-1].reshape(-1), a[1:].reshape(-1))[0, 1]  # consecutive samples of a particle
    assert abs(r) < 5 / sqrt(a[1:].size)
    r2 = np.corrcoef((a[:, :, 0::2] ** 2).reshape(-1), (a[:, :, 1::2] ** 2).reshape(-1))[0, 1]
    # the two members of a pair share their RADIUS: z0^2 + z1^2 = r^2, so their squares are anti-correlated exactly as for true
    # Box-Muller pairs of independent normals - i.e. not at all
    assert abs(r2) < 5 / sqrt(a[:, :, 0::2].size), r2
