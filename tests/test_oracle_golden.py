"""Pin the CPU oracle (oracle/dust_oracle.c) against vectors produced by the reference itself.

Every expected value below was computed by lubaroli/dust (imported in the build container by
tests/golden/make_golden.py); inputs - including every random draw the reference made - are replayed into the oracle.
Tolerance: 1e-5 relative (fp32), the figure BASELINE.json's north_star states, written per assertion.
"""
import numpy as np
import pytest

import oracle
from oracle import Oracle
from helpers import ctrl_noise_of, elemerr, is_adam, relerr, scenario_kwargs

TOL = 1e-5
SVMPC_CASES = ["pend_k1", "pend_k1_close", "pend_k2", "pend_k1_params", "pend_k1_expcost", "pend_k1_ctrlpen", "pend_k1_mean",
               "pend_cfg1", "part_k1_gmm", "part_k2_gmm", "part_k2shared", "part_k1_scalar", "part_k1_near_obst",
               "pend_k1_adam", "part_k1_adam", "pend_k1_f64", "pend_k1_mid_f64", "pend_k2_fixedbw", "part_k2shared_fixedbw",
               # round 5: control-channel noise (particle.py:145-148) and velocity control (particle.py:152-153)
               "part_k1_noisy", "part_k1_velocity", "part_k2_noisy_vel", "part_k1_noisy_zero",
               "pend_k2_minbw",
               "part_k1_fullcov", "part_k2_fullcov"]  # full 2 x 2 a_cov and prior covariance (disco.py:91-98, svgd.py:84-89)  # RBF(minimum_bw=1.4) under the median trick: about half of the per-dimension bandwidths are clamped
K1_F64_CASES = ["pend_k1_f64", "pend_k1_mid_f64"]


def k1_tolerance(theta):
    """K1's reference side is a gpytorch stand-in (third party, parity unpinned) whose squared distance is the fp32
    mean-centred matmul trick |a|^2 + |b|^2 - 2ab: its cancellation noise is ~ eps_f32 * |x/l|^2 on d^2, i.e. up to that
    much RELATIVE noise on every Gram entry, the unit diagonal included (measured: K_ii = 0.99988 in part_k1_gmm).  The
    oracle uses exact differences, so it can only agree with such goldens to that bound."""
    x = np.asarray(theta, np.float64).reshape(theta.shape[0], -1) / np.log(2.0)
    x = x - x.mean(0, keepdims=True)
    return max(TOL, 4 * 6e-8 * float((x * x).sum(1).max()))


def _sig(g, key):
    v = np.asarray(g[key], np.float32).reshape(-1)
    return np.full(int(g["da"]), float(v[0]), np.float32) if v.size == 1 else v


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_actions_rollout_costs(golden, name):
    g = golden(name)
    o = Oracle(**scenario_kwargs(g))
    T, K = g["eps"].shape[:2]
    theta = g["theta0"]
    a_mat = g["a_mat0"]
    for t in range(T):
        for k in range(K):
            actions = o.sample_actions(theta, g["eps"][t, k], _sig(g, "sigma_a"))
            if "a_cov" in g:  # (L eps of a full L is a two-term sum per row: torch's matmul may contract it; one ulp)
                assert elemerr(actions, g["actions"][t, k]) < 1e-6
            else:
                assert np.array_equal(actions, g["actions"][t, k]), "a1: theta + L eps must be bit-exact"
            params = g["params"][t, k] if "params" in g else None
            a_reg = float(g["a_reg"])
            a_pre = 1.0 / _sig(g, "sigma_a") ** 2
            cz = ctrl_noise_of(g, t, k)
            if k == 0:
                costs, states = o.rollout_cost(g["state"][t, k], actions, params, a_reg, a_mat, None, a_pre, want_states=True, ctrl_noise=cz)
                assert elemerr(states, g["states_iter0"][t]) < TOL
                if cz is not None and float(np.abs(g["dyn_std"]).max()) > 0:  # the fixture must notice the noise
                    _, quiet = o.rollout_cost(g["state"][t, k], actions, params, a_reg, a_mat, None, a_pre, want_states=True)
                    assert elemerr(quiet, g["states_iter0"][t]) > 1e-3
            else:
                costs = o.rollout_cost(g["state"][t, k], actions, params, a_reg, a_mat, None, a_pre, ctrl_noise=cz)
            assert elemerr(costs, g["costs"][t, k]) < TOL, name
            # a6 side effects (MultiDISCO.forward): a_mat += sum_s omega eps ; a_mix
            _, a_mat, a_mix = o.disco_weights(g["costs"][t, k], actions, np.zeros(o.D), float(g["temperature"]), a_mat)
            # omega-weighted sums of signed noise cancel (Particle: a handful of samples carry all the weight): 1e-4 element-wise
            # (full covariance: the actions themselves agree to an ulp only - above - and the sum cancels: 1e-4 as well)
            assert elemerr(a_mat, g["omega_amat"][t, k]) < (TOL if (name != "part_k1_adam" and "a_cov" not in g) else 1e-4)
            assert relerr(a_mix, g["a_mix"][t, k], floor=1e-30) < 1e-4  # softmax of O(1e3) logits: ulp(cost) amplification
            a_mat = g["omega_amat"][t, k]
            theta = g["theta_after"][t, k]
        theta = g["tick_theta_rolled"][t]


def _prior_at(g, t, theta):
    """Prior (means, mixture weights) in force at tick t.

    Reference quirk that defines "correct": SVMPC.update_prior (svmpc.py:160-170) builds the new GMM from
    `self.theta.detach()` (svgd.py:87), which ALIASES theta's storage; the optimizer then updates theta in place, so from
    the second tick on the prior means are always the CURRENT particles."""
    if t == 0:
        return g["mu0"], g["mix0"]
    mix = g["tick_p_weights"][t - 1] if int(g["weighted_prior"]) else np.ones(int(g["N"]), np.float32)
    return theta, mix


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_score_phi_update(golden, name):
    """a9-a11 + a8, fed with the reference's own costs/actions (stage-wise parity, SURVEY 'tolerance amplification')."""
    g = golden(name)
    o = Oracle(**scenario_kwargs(g))
    T, K = g["eps"].shape[:2]
    kind = str(g["kernel_kind"])
    theta = g["theta0"]
    for t in range(T):
        for k in range(K):
            mu, mix = _prior_at(g, t, theta)
            gl, gp, sc = o.score(theta, mu, mix, _sig(g, "sigma_p"), g["costs"][t, k], g["actions"][t, k], float(g["alpha"]),
                                 _sig(g, "sigma_a"))
            assert elemerr(gp, g["grad_pri"][t, k]) < TOL
            if kind == "K1":
                phi = o.phi_k1(theta, sc, variant=0)
                tol = k1_tolerance(theta)
            elif kind == "K2":
                phi, hk = o.phi_k2(theta, sc, indep=True, bandwidth=float(g["k2_bandwidth"]) if "k2_bandwidth" in g else -1.0,
                                   minimum_bw=float(g["k2_minimum_bw"]) if "k2_minimum_bw" in g else 1e-5)
                tol = TOL
                if "k2_minimum_bw" in g and (t, k) == (0, 0):  # the fixture must have clamped AND free dimensions
                    assert 2 <= int((hk == np.float32(g["k2_minimum_bw"])).sum()) <= hk.size - 2
            else:
                phi, _ = o.phi_k2(theta, sc, indep=False, bandwidth=float(g["k2_bandwidth"]) if "k2_bandwidth" in g else -1.0)
                tol = TOL
            assert elemerr(phi, g["phi"][t, k]) < tol, (name, t, k)
            if is_adam(name):  # the reference's class default (svgd.py:115); its state restarts at every roll (svmpc.py:142-158)
                if k == 0:
                    m, v = np.zeros_like(theta), np.zeros_like(theta)
                th1, m, v = o.adam(theta, g["phi"][t, k], m, v, k + 1, float(g["lr"]))
                assert elemerr(th1, g["theta_after"][t, k]) < 2e-6, (name, t, k)
            else:
                th1 = o.sgd(theta, g["phi"][t, k], float(g["lr"]))
                assert elemerr(th1, g["theta_after"][t, k]) < 1e-6
            theta = g["theta_after"][t, k]
        theta = g["tick_theta_rolled"][t]


@pytest.mark.parametrize("name", K1_F64_CASES)
def test_k1_branch_vs_float64_reference(golden, name):
    """K1 (gpytorch RBFKernel semantics; third party, absent) against the SAME reference call evaluated in float64 on the
    recorded fp32 inputs: the fp32 stand-in's matmul-trick distance carries cancellation noise (k1_tolerance), the float64 run
    does not - so here the oracle meets 1e-5, element-wise, on particle sets close enough for a non-trivial Gram matrix."""
    g = golden(name)
    o = Oracle(**scenario_kwargs(g))
    T, K = g["eps"].shape[:2]
    worst_fp32, interacting = 0.0, 0
    for t in range(T):
        for k in range(K):
            phi, gram = o.phi_k1(g["theta_in"][t, k], g["score"][t, k], variant=0, want_gram=True)
            interacting += int((gram - np.diag(np.diag(gram))).max() > 1e-2)
            assert elemerr(phi, g["phi_f64"][t, k]) < TOL, (name, t, k)
            worst_fp32 = max(worst_fp32, elemerr(g["phi"][t, k], g["phi_f64"][t, k]))
    assert interacting >= 2, "the fixture must have interacting particles (a Gram matrix that is not the identity)"
    assert worst_fp32 < 5e-3  # (the fp32 stand-in itself is only this close to its own float64 evaluation)


@pytest.mark.parametrize("name", SVMPC_CASES)
def test_forward(golden, name):
    """a12: weights, argmax, roll, prior refresh."""
    g = golden(name)
    o = Oracle(**scenario_kwargs(g))
    T, K = g["eps"].shape[:2]
    lik = oracle.LIK_EXP_UTILITY if str(g["lik_kind"]) == "ExponentiatedUtility" else oracle.LIK_EXPECTED_COST
    roll = oracle.ROLL_REPEAT if str(g["roll_strategy"]) == "repeat" else oracle.ROLL_MEAN
    for t in range(T):
        mu, mix = _prior_at(g, t, g["theta_after"][t, K - 1])
        r = o.forward(g["costs"][t, K - 1], g["theta_after"][t, K - 1], mu, mix, _sig(g, "sigma_p"), float(g["alpha"]), lik,
                      bool(int(g["weighted_prior"])), roll)
        assert elemerr(r["log_l"], g["tick_log_l"][t]) < TOL
        assert elemerr(r["log_p"], g["tick_log_p"][t]) < TOL
        assert r["i_star"] == int(np.argmax(g["tick_p_weights"][t]))
        assert relerr(r["p_weights"], g["tick_p_weights"][t]) < 2e-3  # exp of O(1e3) log-weights: 1 ulp of log_l = 2e-4 rel
        assert np.array_equal(r["a_seq"], g["tick_a_seq"][t])
        assert relerr(r["theta"], g["tick_theta_rolled"][t]) < 1e-6
        assert np.array_equal(r["mu"], r["theta"])
        assert relerr(r["mu"], g["tick_prior_means"][t]) < 1e-6
        pm = r["mix"] / r["mix"].sum()
        assert relerr(pm, g["tick_prior_probs"][t]) < 2e-3


def test_disco_mppi(golden):
    """MultiDISCO.forward with internally drawn noise and ctrl_penalty != 1, then step() (a6, a14)."""
    g = golden("disco_mppi")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    o = Oracle(model="pendulum", N=N, S=S, M=1, H=H)
    sig = np.array([float(g["sigma_a"])], np.float32)
    actions = o.sample_actions(g["a_mat0"], g["z"], sig)  # actions = a_mat + L z (disco.py:155-160)
    assert relerr(actions, g["actions"][0]) < 1e-7
    costs, states = o.rollout_cost(g["state"], actions, None, float(g["a_reg"]), g["a_mat0"], None, 1.0 / sig ** 2, want_states=True)
    assert relerr(states, g["states"]) < TOL
    assert relerr(costs, g["costs"]) < TOL
    omega, a_mat, a_mix = o.disco_weights(g["costs"], actions, g["a_mat0"], float(g["temperature"]), g["a_mat0"])
    assert relerr(omega, g["omega"]) < 1e-4
    assert relerr(a_mat, g["a_mat1"]) < TOL
    assert relerr(a_mix, g["a_mix"]) < 1e-4
    for strat in ("argmax", "average"):
        nxt, a_seq, am = o.disco_step(g["a_mat1"], g["a_mix"], strat, 2, -2.0, 2.0)
        assert relerr(nxt, g["step_%s_actions" % strat]) < 1e-6
        assert relerr(a_seq, g["step_%s_a_seq" % strat]) < 1e-6
        assert relerr(am, g["step_%s_a_mat" % strat]) < 1e-6
    nxt, a_seq, _ = o.disco_step(g["a_mat1"], g["a_mix"], "external", 1, -2.0, 2.0, ext=g["step_external_in"])
    assert np.array_equal(nxt, g["step_external_actions"])
    assert np.array_equal(a_seq, g["step_external_a_seq"])


def test_maps_and_collisions(golden):
    g = golden("maps")
    ref = np.unpackbits(g["map_grid_4x4_w2p1"])[: 220 * 220].reshape(220, 220)
    m = oracle.grid_4x4_map(2.1)
    assert np.array_equal(ref, m.astype(np.uint8))
    assert int(m.sum()) == int(g["n_occupied_grid_4x4"]) == 8620
    o = Oracle(model="particle", uncertain_params=("mass",), grid=m)
    assert np.array_equal(o.get_collisions(g["points"]), g["collisions"])


@pytest.mark.parametrize("name", ["mpf_pend", "mpf_part_log"])
def test_mpf(golden, name):
    g = golden(name)
    kind = str(g["model_kind"])
    up = ("length", "mass") if kind == "pendulum" else ("mass",)
    o = Oracle(model=kind, uncertain_params=up, mass=2.0 if kind == "particle" else 1.0)
    bw, ls = float(g["bw"]), bool(int(g["log_space"]))
    phi0 = o.mpf_phi(g["x0"], g["x0"], bw, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), ls, bw)
    assert relerr(phi0, g["phi0"]) < TOL
    x, pm, pbw, gn = o.mpf_optimize(g["x0"], g["x0"], bw, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), ls, bw,
                                    float(g["lr"]), int(g["n_steps"]))
    assert relerr(x, g["x_final"]) < TOL
    assert relerr(gn, g["grad_norms"]) < TOL
    assert relerr(pm, g["prior_means"]) < TOL
    x2, pm2, _, gn2 = o.mpf_optimize(x, pm, pbw, g["obs1"], g["action2"], g["obs2"], float(g["obs_std"]), ls, bw, float(g["lr"]),
                                     int(g["n_steps"]))
    assert relerr(x2, g["x_final2"]) < TOL
    # (y - pred) cancels ~100x here (|y - pred| ~ 6e-3 on |pred| ~ 0.5), so a 1-ulp difference between libm expf and
    # torch's vectorised exp in mass = exp(x) shows up at ~1e-4 in the first steps' gradient norms; x itself agrees to 1e-5.
    assert relerr(gn2, g["grad_norms2"]) < 2e-4
    assert relerr(Oracle.gmm_log_prob(g["probe"], pm2, bw), g["probe_log_prob"]) < TOL


@pytest.mark.parametrize("name", ["mpf_pend_adam", "mpf_part_log_adam"])
def test_mpf_adam(golden, name):
    """MPF with the reference's class-default optimiser (torch.optim.Adam, svgd.py:115), two filter updates: the optimiser is built
    once (mpf.py:24), so its moments and step count carry over from the first optimize() to the second."""
    g = golden(name)
    assert str(g["optimizer"]) == "Adam"
    kind = str(g["model_kind"])
    up = ("length", "mass") if kind == "pendulum" else ("mass",)
    o = Oracle(model=kind, uncertain_params=up, mass=2.0 if kind == "particle" else 1.0)
    bw, ls, lr, n = float(g["bw"]), bool(int(g["log_space"])), float(g["lr"]), int(g["n_steps"])
    x, pm, pbw, gn, m, v, st = o.mpf_optimize_adam(g["x0"], g["x0"], bw, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), ls, bw, lr, n)
    assert st == n
    assert relerr(x, g["x_final"]) < TOL
    assert relerr(gn, g["grad_norms"]) < 2e-4  # (see test_mpf: the first gradient norms carry the libm-vs-torch exp ulp)
    x2, pm2, _, gn2, _, _, st2 = o.mpf_optimize_adam(x, pm, pbw, g["obs1"], g["action2"], g["obs2"], float(g["obs_std"]), ls, bw, lr, n,
                                                    m=m, v=v, step=st)
    assert st2 == 2 * n
    assert relerr(x2, g["x_final2"]) < TOL
    assert relerr(gn2, g["grad_norms2"]) < 2e-4
    # restarting the optimiser state at the second call (what SVMPC does at every roll) must NOT reproduce the reference here
    x2r, *_ = o.mpf_optimize_adam(x, pm, pbw, g["obs1"], g["action2"], g["obs2"], float(g["obs_std"]), ls, bw, lr, n)
    assert relerr(x2r, g["x_final2"]) > 10 * TOL


def _noisy_actions(action, dyn_std, z):
    """acts = action + dyn_std * z in fp32, one draw per phi() call (particle.py:145-148 reached through likelihoods.py:30-46)."""
    return (np.asarray(action, np.float32)[None, :] + (np.asarray(dyn_std, np.float32)[None, :] * np.asarray(z, np.float32))).astype(np.float32)


def test_mpf_control_noise(golden):
    """MPF over Particle(deterministic=False): every phi() call of the reference's filter draws ONE control-noise vector (`acts` is the
    bare past action at likelihoods.py:44) - the one-step prediction and its Jacobian use action + dyn_std * z for all particles.
    The recorded draws are replayed step by step through the oracle's phi + SGD."""
    g = golden("mpf_part_noisy")
    o = Oracle(model="particle", uncertain_params=("mass",), mass=2.0)
    bw, lr, n, std = float(g["bw"]), float(g["lr"]), int(g["n_steps"]), g["dyn_std"]
    a0 = _noisy_actions(g["action"], std, g["phi0_noise"][None])[0]
    assert relerr(o.mpf_phi(g["x0"], g["x0"], bw, g["obs0"], a0, g["obs1"], float(g["obs_std"]), True, bw), g["phi0"]) < TOL
    assert relerr(o.mpf_phi(g["x0"], g["x0"], bw, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), True, bw), g["phi0"]) > 1e-3
    x = g["x0"].copy()
    for past, act, obs, zs, want_x, want_gn in ((g["obs0"], g["action"], g["obs1"], g["noise1"], g["x_final"], g["grad_norms"]),
                                                (g["obs1"], g["action2"], g["obs2"], g["noise2"], g["x_final2"], g["grad_norms2"])):
        gn = []
        for a_eff in _noisy_actions(act, std, zs):
            phi = o.mpf_phi(x, x, bw, past, a_eff, obs, float(g["obs_std"]), True, bw)
            gn.append(float(np.sqrt((phi.astype(np.float64) ** 2).sum())))
            x = o.sgd(x, phi, lr)
        assert relerr(x, want_x) < TOL
        assert relerr(np.asarray(gn), want_gn) < 2e-4


def test_unscented_transform_costs_vs_reference(golden):
    """sigma-point rollouts.  The oracle's restatement (incl. the reference's (sigma, step) weight pattern,
    disco.py:314-316) against MultiDISCO(params_sampling=MerweScaledUTF(n=2, alpha=0.5)).forward of the reference."""
    from dust_amd.utils.utf import MerweScaledUTF
    from oracle import Oracle

    g = golden("disco_ut")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    tf = MerweScaledUTF(n=2, alpha=0.5)
    assert np.allclose(tf.loc_weights.numpy(), g["loc_weights"], rtol=1e-6)
    sp = tf.compute_sigma_points(g["dyn_mean"], np.diag(g["dyn_var"]))
    assert np.allclose(sp.numpy(), g["sigma_points"], rtol=1e-6)
    o = Oracle(model="pendulum", N=N, S=S, M=tf.pts, H=H, uncertain_params=("length", "mass"))
    spT = np.ascontiguousarray(g["sigma_points"].T)
    assert relerr(o.rollout_cost_ut(g["state"], g["actions"], spT, g["loc_weights"]), g["costs"]) < 1e-5
    assert relerr(o.rollout_cost_ut(g["state"], g["ext_actions"], spT, g["loc_weights"]), g["costs_ext"]) < 1e-5


@pytest.mark.parametrize("name", ["mpf_bwvec", "mpf_bwiqr"])
def test_mpf_initial_prior_from_bw_silverman(golden, name):
    """MPF(bw=None) (mpf.py:29-38): the first prior's covariance is diag(bw_silverman(columns)^2) - one bandwidth per parameter
    when `_select_sigma` (svgd.py:10-25) takes its per-column std branch (`mpf_bwvec`), a scalar on the pooled-IQR branch
    (`mpf_bwiqr`).  The host mirror of bw_silverman, the oracle's prior density and phi under that prior, and two optimize() calls
    (the prior is isotropic again after the first: update_prior(bw), mpf.py:85) against the reference's own MPF."""
    from dust_amd.inference.svgd import bw_silverman
    from oracle import Oracle

    g = golden(name)
    P = int(g["P"])
    bw0 = np.asarray(bw_silverman(g["x0"], 1.0), np.float32).reshape(-1)
    assert bw0.size == g["bw_init"].size and relerr(bw0, g["bw_init"]) < 1e-6
    bwv = np.broadcast_to(bw0, (P,)).astype(np.float32)
    assert relerr(bwv ** 2, g["prior_cov_diag"]) < 1e-6 and float(g["prior_cov_offdiag_max"]) == 0.0
    o = Oracle(model="pendulum", uncertain_params=("length", "mass"))
    assert relerr(Oracle.gmm_log_prob_v(g["probe"], g["x0"], bwv), g["probe_log_prob0"]) < TOL
    bw, lr, n = float(g["bw_opt"]), float(g["lr"]), int(g["n_steps"])
    phi0 = o.mpf_phi_v(g["x0"], g["x0"], bwv, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), False, bw)
    assert relerr(phi0, g["phi0"]) < TOL
    x, pm, bv, gn = o.mpf_optimize_v(g["x0"], g["x0"], bwv, g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), False, bw, lr, n)
    # the prior attracts with stiffness 1 / bw_min^2 (1 450 at bw_0 = 0.026): an SGD step multiplies a difference in x by up to
    # 1 + lr / bw_min^2 = 2.45, so n steps carry the fp32 noise of phi (TOL) to TOL * 2.45^n on x
    tol_x = TOL * (1.0 + lr / float(bwv.min()) ** 2) ** n
    assert relerr(x, g["x_final"]) < tol_x and relerr(gn, g["grad_norms"]) < TOL
    assert np.all(bv == np.float32(bw))
    assert relerr(Oracle.gmm_log_prob(g["probe"], pm, bw), g["probe_log_prob1"]) < TOL
    x2, _, _, gn2 = o.mpf_optimize(x, pm, bw, g["obs1"], g["action2"], g["obs2"], float(g["obs_std"]), False, bw, lr, n)
    assert relerr(x2, g["x_final2"]) < 2 * tol_x and relerr(gn2, g["grad_norms2"]) < 2e-4
    if name == "mpf_bwvec":  # a scalar prior bandwidth must NOT reproduce the reference here
        bad = o.mpf_phi(g["x0"], g["x0"], float(bwv[0]), g["obs0"], g["action"], g["obs1"], float(g["obs_std"]), False, bw)
        assert relerr(bad, g["phi0"]) > 100 * TOL


@pytest.mark.parametrize("name", ["skid_nominal", "skid_params", "skid_params_log"])
def test_skid_steer_rollouts_vs_reference(golden, name):
    """SkidSteerRobot.step (skid_steer_robot.py:73-122) under MultiDISCO._rollout / _compute_cost with a quadratic cost:
    the oracle's restatement against the reference's own MultiDISCO.forward - costs and every state of every rollout, nominal
    parameters, sampled (x_icr, wheel_radius) and log-space samples; ~20 % of the actions are clamped by the step."""
    from oracle import Oracle

    g = golden(name)
    up = ("x_icr", "wheel_radius") if "params" in g else ()
    costs, states = Oracle.skid_rollout_cost(g["state"], g["ext_actions"], params=g["params"] if up else None, uncertain_params=up, dt=float(g["dt"]),
                                             goal=g["goal"], w_state=g["w_state"], w_term=g["w_term"], w_ctrl=g["w_ctrl"],
                                             log_space=bool(int(g["params_log_space"])), want_states=True)
    assert float(g["clamped_fraction"]) > 0.1
    assert relerr(costs, g["costs"]) < TOL
    assert np.abs(states - g["states"]).max() < 1e-5 * max(1.0, np.abs(g["states"]).max())
