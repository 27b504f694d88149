"""The reference's class API (MultiDISCO / SVMPC / likelihoods / MPF / get_gmm) on top of the C ABI, replaying golden
scenarios through the SAME call sequence the reference's simulation loop uses (simulations.py:104-138)."""
import copy

import numpy as np
import pytest
import torch

from helpers import RecordedDraws, elemerr, relerr
from test_host_mirror_cpu import PARTICLE_ENV
from test_oracle_golden import k1_tolerance

pytestmark = pytest.mark.gpu


def inst_cost(states, controls=None, n_pol=1, debug=None):  # the demo's own, un-tagged cost functions
    theta, theta_d = states.chunk(2, dim=1)
    return 50.0 * (theta.cos() - 1) ** 2 + 1.0 * theta_d ** 2


def term_cost(states, n_pol=1, debug=None):
    return inst_cost(states).squeeze()


def build_pendulum(g, kernel):
    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import SVMPC, ExponentiatedUtility, get_gmm
    from dust_amd.models import PendulumModel

    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    model = PendulumModel()
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=float(g["temperature"]),
                      a_cov=float(g["sigma_a"]) ** 2 * torch.eye(1), inst_cost_fn=inst_cost, term_cost_fn=term_cost, params_sampling=None)
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), float(g["sigma_p"]) ** 2 * torch.eye(1))
    lik = ExponentiatedUtility(alpha=float(g["alpha"]), n_samples=S, controller=ctrl, model=model)
    sv = SVMPC(init_particles=torch.tensor(g["theta0"]), prior=prior, likelihood=lik, kernel=kernel, n_particles=N, bw_scale=1.0,
               n_steps=1, optimizer_class=torch.optim.SGD, lr=float(g["lr"]))
    return model, ctrl, lik, sv


@pytest.mark.parametrize("name,kernel_name", [("pend_k1", "rbf"), ("pend_k2", "message_passing")])
def test_simulation_loop_call_sequence(golden, name, kernel_name):
    from dust_amd.kernels import RBF, RBFKernel, iid_mp

    g = golden(name)
    kernel = RBFKernel() if kernel_name == "rbf" else iid_mp(base_kernel=RBF(bandwidth=-1), ctrl_dim=1, indep_controls=True)
    model, ctrl, lik, sv = build_pendulum(g, kernel)
    T, K = g["eps"].shape[:2]
    for t in range(T):
        state = torch.tensor(g["state"][t, 0]).unsqueeze(0)
        sv.optimize(state, None, n_steps=K, eps=g["eps"][t])           # sim_svmpc.optimize(state, dyn_dist)
        scale = np.abs(g["theta_after"][t, K - 1]).max()
        assert np.abs(sv.theta.numpy() - g["theta_after"][t, K - 1]).max() / scale < 2e-3
        a_seq, p_w = sv.forward(state, None)                           # sim_svmpc.forward(state, dyn_dist)
        assert a_seq.shape == (int(g["H"]), 1) and abs(float(p_w.sum()) - 1) < 5e-4
        assert int(p_w.argmax()) == int(np.argmax(g["tick_p_weights"][t]))
        sv.theta = torch.tensor(g["tick_theta_rolled"][t])             # re-sync with the reference (stage-local checks)
    assert ctrl.a_mat.shape == (int(g["N"]), int(g["H"]), 1)


def test_phi_hook_and_likelihood_api(golden):
    from dust_amd.kernels import RBFKernel

    g = golden("pend_k1")
    model, ctrl, lik, sv = build_pendulum(g, RBFKernel())
    state = torch.tensor(g["state"][0, 0])
    costs, actions = lik.sample(torch.tensor(g["theta0"]), state, None, eps=g["eps"][0, 0])
    assert np.array_equal(actions.numpy(), g["actions"][0, 0])
    assert elemerr(costs.numpy(), g["costs"][0, 0]) < 1e-5
    assert relerr(lik.log_prob(costs).numpy(), lik.log_prob(torch.tensor(g["costs"][0, 0])).numpy()) < 1e-5

    def log_p(theta):  # the reference's phi() takes exactly such a callable (svmpc.py:88-90)
        return None, torch.tensor(g["costs"][0, 0]), torch.tensor(g["actions"][0, 0])

    phi = sv.phi(log_p, None, None)
    assert elemerr(phi.numpy(), g["phi"][0, 0]) < k1_tolerance(g["theta0"])


def test_deepcopy_isolates_device_state(golden):
    from dust_amd.kernels import RBFKernel

    g = golden("pend_k1")
    model, ctrl, lik, sv = build_pendulum(g, RBFKernel())
    state = torch.tensor(g["state"][0, 0])
    sv.optimize(state, None, n_steps=1, eps=g["eps"][0, :1])
    sv2 = copy.deepcopy(sv)                                            # simulations.py:62,78 / particle_example.py:166-175
    assert sv2.likelihood.controller is not ctrl and sv2.likelihood.controller._ctx is not ctrl._ctx
    t_before = sv.theta.clone()
    sv2.optimize(state, None, n_steps=1, eps=g["eps"][0, 1:2])
    assert torch.equal(sv.theta, t_before) and not torch.equal(sv2.theta, t_before)
    sv.optimize(state, None, n_steps=1, eps=g["eps"][0, 1:2])
    assert torch.equal(sv.theta, sv2.theta)                            # same inputs -> bitwise same result (no atomics)


def test_multidisco_forward_and_step(golden):
    from dust_amd.controllers import MultiDISCO
    from dust_amd.models import PendulumModel

    g = golden("disco_mppi")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    model = PendulumModel()
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=0.7, ctrl_penalty=0.4, a_cov=1.5 ** 2 * torch.eye(1),
                      inst_cost_fn=inst_cost, term_cost_fn=term_cost, params_sampling=None)
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    costs, states, actions, omega, plp = ctrl.forward(torch.tensor(g["state"]), model)  # own (Philox) noise
    assert costs.shape == (S, N) and states.shape == (1, S, N, H + 1, 2) and actions.shape == (1, S, N, H, 1) and plp is None
    assert torch.allclose(omega.sum(0), torch.ones(N), atol=1e-5)
    a = ctrl.step(strategy="average")
    assert a.shape == (1, 1) and float(a.abs().max()) <= 2.0
    with pytest.raises(ValueError):
        ctrl.step(strategy="nonsense")
    with pytest.raises(NotImplementedError):
        MultiDISCO(model.observation_space, model.action_space, H, N, S, inst_cost_fn=lambda s, c=None, **k: s.sum(1, keepdim=True),
                   term_cost_fn=term_cost, params_sampling=None).forward(torch.tensor(g["state"]), model)


def test_particle_dual_inference_loop(golden):
    """particle_example.py:150-207 call sequence with MPF coupled in: controller draws its dynamics samples from mpf.prior."""
    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import MPF, SVMPC, ExponentiatedUtility, GaussianLikelihood, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import Particle

    torch.manual_seed(0)
    N, H, S, M = 8, 12, 16, 4
    model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    system = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=1.0, a_cov=25.0 * torch.eye(2), params_sampling=True,
                      params_samples=M, params_log_space=True, inst_cost_fn=model.default_inst_cost, term_cost_fn=model.default_term_cost)
    prior = get_gmm(torch.randn(N, H, 2), torch.ones(N), 25.0 * torch.eye(2))
    lik = ExponentiatedUtility(1.0, controller=ctrl, model=model, n_samples=S)
    sv = SVMPC(init_particles=prior.sample([N]), prior=prior, likelihood=lik, kernel=RBFKernel(), n_particles=N, n_steps=1,
               optimizer_class=torch.optim.SGD, lr=100.0, weighted_prior=True)
    state = torch.tensor([-9.0, -9.0, 0.0, 0.0])
    x0 = torch.distributions.Normal(2.0, 0.1).sample([16, 1]).clamp(min=1e-6).log()
    mpf = MPF(init_particles=x0, likelihood=GaussianLikelihood(state, 0.1, model, log_space=True), optimizer_class=torch.optim.SGD, lr=0.01,
              bw=0.1, bw_scale=1.0)
    for step in range(3):
        sv.optimize(state, mpf.prior)
        a_seq, _ = sv.forward(state, mpf.prior)
        action = a_seq[0]
        state = system.step(state, action)
        grads, bw = mpf.optimize(action, state, bw=0.5, n_steps=5)
        assert grads.shape == (5,) and torch.isfinite(grads).all() and torch.isfinite(sv.theta).all()
    assert mpf.x.shape == (16, 1) and mpf.prior.sample([4]).shape == (4, 1)


def test_closed_loop_drivers():
    """the reference's closed-loop entry points (simulations.py) on the device backend - the particle
    episode with the mass-change event and the dynamics filter, and the pendulum simulation's result frame."""
    import importlib.util
    import os

    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import MPF, GaussianLikelihood, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import PendulumModel
    from dust_amd.utils.simulations import run_pendulum_simulation

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("particle_example", os.path.join(root, "examples", "particle_example.py"))
    pe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pe)
    costs = pe.main(["--steps", "24", "--seed", "1"])
    assert len(costs) == 1 and np.isfinite(costs[0]) and costs[0] > 0

    torch.manual_seed(0)
    N, H, S = 4, 10, 32
    env_model = PendulumModel()
    prior = get_gmm(torch.randn(N, H, 1), torch.ones(N), 4.0 * torch.eye(1))
    init_policies = prior.sample([N])
    dyn_prior = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])), 1)
    ctrl = MultiDISCO(env_model.observation_space, env_model.action_space, H, N, S, temperature=1.0, a_cov=4.0 * torch.eye(1),
                      inst_cost_fn=inst_cost, term_cost_fn=term_cost, params_sampling=True, params_samples=3)
    init_state = torch.tensor([3.0, 0.0])
    lik = GaussianLikelihood(initial_obs=init_state, obs_std=0.1, model=PendulumModel(uncertain_params=("length", "mass")), log_space=False)
    mpf = MPF(init_particles=dyn_prior.sample([16]), likelihood=lik, optimizer_class=torch.optim.SGD, lr=1e-3, bw=0.1)
    df = run_pendulum_simulation(
        init_state, init_policies, dict(uncertain_params=("length", "mass")), mpf.prior, [dict(length=0.9, mass=1.1)], ctrl,
        use_exact_model=False, use_svmpc=True,
        svmpc_kwargs=dict(init_particles=init_policies, prior=prior, kernel=RBFKernel(), n_particles=N, bw_scale=1.0, n_steps=1,
                          optimizer_class=torch.optim.SGD, lr=2.0),
        lik_kwargs=dict(alpha=1.0, n_samples=S), mpf=mpf, mpf_bw=None, mpf_steps=5, episodes=1, steps=6, warm_up=1)
    want = {"Cost", "Position", "Speed", "Actions", "Timestep", "Iteration", "DynParticles", "DynBandwidths", "PolParticles", "Weights",
            "ExpParams", "AvgCumCost"}
    assert want <= set(df.columns) and len(df) == 6
    assert np.isfinite(df["Cost"].to_numpy()).all() and np.isfinite(df["Position"].to_numpy()).all()
    assert abs(sum(df["Weights"].iloc[3]) - 1.0) < 1e-3


def test_unscented_transform_disco_vs_reference(golden):
    """SURVEY 8(f).3, the reference's "DISCO" case: MultiDISCO(params_sampling=MerweScaledUTF) - sigma-point rollouts of the
    dynamics distribution, UT-weighted costs, MPPI update and step("average") - against the reference's own outputs."""
    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.models import PendulumModel
    from dust_amd.utils.utf import MerweScaledUTF

    g = golden("disco_ut")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    tf = MerweScaledUTF(n=2, alpha=0.5)
    dyn = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.7]), torch.tensor([1.3, 1.2])), 1)
    model = PendulumModel(length=dyn.mean[0], mass=dyn.mean[1], uncertain_params=("length", "mass"))
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=0.8, a_cov=1.2 ** 2 * torch.eye(1),
                      inst_cost_fn=inst_cost, term_cost_fn=term_cost, params_sampling=tf, params_log_space=False)
    assert ctrl.n_params == 1  # disco.py:128: the sigma points are internal to the rollouts
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    state = torch.tensor(g["state"])
    # the reference's recorded draws as external actions: a_mat + L z (disco.py:229-233)
    costs, states, actions, omega, plp = ctrl.forward(state, model, dyn, ext_actions=torch.tensor(g["actions"]))
    assert elemerr(costs.numpy(), g["costs"]) < 1e-5
    # costs ~ 1.4e3: one fp32 ulp of a cost (1.2e-4) is 1.5e-4 on exp(-cost / 0.8) - the weights carry that amplification
    wtol = 8 * float(np.spacing(np.float32(np.abs(g["costs"]).max()))) / 0.8
    assert relerr(omega.numpy(), g["omega"]) < wtol
    assert relerr(plp.numpy(), g["params_log_p"]) < 1e-5
    assert relerr(ctrl.a_mix.numpy(), g["a_mix"]) < wtol
    # second call, external actions around the updated plan
    ctrl.a_mat = torch.tensor(g["a_mat1"])
    costs2, _, _, omega2, _ = ctrl.forward(state, model, dyn, ext_actions=torch.tensor(g["ext_actions"]))
    assert elemerr(costs2.numpy(), g["costs_ext"]) < 1e-5 and relerr(omega2.numpy(), g["omega_ext"]) < wtol
    # own noise + step("average"): shapes / bounds (the draws differ from torch's)
    ctrl.forward(state, model, dyn)
    a = ctrl.step(strategy="average")
    assert a.shape == (1, 1) and float(a.abs().max()) <= 2.0
    c2 = copy.deepcopy(ctrl)  # the weights travel with the clone
    c3, *_ = c2.forward(state, model, dyn, ext_actions=torch.tensor(g["ext_actions"]))
    assert np.isfinite(c3.numpy()).all()


def test_forward_pieces_get_weights_roll_update_prior(golden):
    """SVMPC.get_weights / roll / update_prior one by one (svmpc.py:128-170), the refreshed `prior` object, roll strategies
    incl. 'resample' (host-drawn prior sample, torch's RNG as in the reference), forward(steps=-2), and likelihood.sample(theta)
    leaving the optimiser's particles alone (likelihoods.py:81-101 takes theta as an argument)."""
    from dust_amd.kernels import RBFKernel

    g = golden("pend_k1")
    N, H = int(g["N"]), int(g["H"])
    model, ctrl, lik, sv = build_pendulum(g, RBFKernel())
    state = torch.tensor(g["state"][0, 0])
    sv.optimize(state, None, n_steps=2, eps=g["eps"][0][:2])
    th = sv.theta.clone()
    # get_weights: weights only - theta and the prior stay
    w = sv.get_weights(state, None)
    assert torch.equal(sv.theta, th)
    assert relerr(sv.prior.component_distribution.base_dist.loc.numpy(), g["mu0"]) < 1e-7
    # likelihood.sample with another theta: the optimiser's particles are untouched
    other = th + 0.5
    lik.sample(other, state, None, eps=g["eps"][0, 0])
    assert torch.equal(sv.theta, th)
    # roll == torch semantics (circular shift, then the last row per strategy)
    for steps, strat in ((-1, "repeat"), (-2, "repeat"), (-1, "mean"), (-3, "mean")):
        sv.theta = th
        sv.roll(steps, strat)
        ref = th.roll(steps, dims=-2)
        ref[..., -1, :] = ref[..., -2, :] if strat == "repeat" else ref.mean(dim=-2)
        assert relerr(sv.theta.numpy(), ref.numpy()) < 1e-6, (steps, strat)
    sv.theta = th
    torch.manual_seed(5)
    sv.roll(-1, "resample")
    torch.manual_seed(5)
    ref = th.roll(-1, dims=-2)
    ref[..., -1, :] = sv.prior.sample([N])[..., -1, :]
    assert relerr(sv.theta.numpy(), ref.numpy()) < 1e-6
    with pytest.raises(ValueError):
        sv.roll(-1, "nope")
    # update_prior: means alias the particles, mixture = ones (weighted_prior False)
    sv.theta = th
    sv.update_prior(w)
    p = sv.prior
    assert relerr(p.component_distribution.base_dist.loc.numpy(), th.numpy()) < 1e-7
    assert relerr(p.mixture_distribution.probs.numpy(), np.full(N, 1.0 / N)) < 1e-6
    # forward(steps=-2): same weights / action sequence as get_weights (same costs, same prior), particles rolled by two
    w = sv.get_weights(state, None)
    a_seq, pw = sv.forward(state, None, steps=-2)
    assert relerr(pw.numpy(), w.numpy()) < 1e-5
    assert torch.equal(a_seq, th[int(w.argmax())])
    ref = th.roll(-2, dims=-2)
    ref[..., -1, :] = ref[..., -2, :]
    assert relerr(sv.theta.numpy(), ref.numpy()) < 1e-6
    assert relerr(sv.prior.component_distribution.base_dist.loc.numpy(), ref.numpy()) < 1e-7  # refreshed prior object
    assert isinstance(sv.optimizer, torch.optim.SGD)
    # a stand-alone roll() AFTER forward(): the prior keeps the pre-roll particles as its means until update_prior() is called
    # (svmpc.py:142 rebinds theta to a new tensor; the GMM of svgd.py:87 keeps the old storage) - on the device the prior aliased
    # theta at this point, so the roll must not drag the means along
    from dust_amd.inference import get_gmm

    pre = sv.theta.clone()
    sv.roll(-1, "repeat")
    rolled = pre.roll(-1, dims=-2)
    rolled[..., -1, :] = rolled[..., -2, :]
    assert relerr(sv.theta.numpy(), rolled.numpy()) < 1e-6
    w2 = sv.get_weights(state, None)
    prior_ref = get_gmm(pre, torch.ones(N), float(g["sigma_p"]) ** 2 * torch.eye(1))
    lw_ref = lik.log_prob() + prior_ref.log_prob(rolled)
    assert np.abs(w2.numpy() - torch.softmax(lw_ref, dim=0).numpy()).max() < 1e-4
    sv._prior_stale = True
    assert relerr(sv.prior.component_distribution.base_dist.loc.numpy(), pre.numpy()) < 1e-7
    sv.update_prior()
    assert relerr(sv.prior.component_distribution.base_dist.loc.numpy(), rolled.numpy()) < 1e-7


def test_resample_strategy_ticks(golden):
    """roll_strategy='resample' through forward() and tick(): the last action row is the last row of a fresh prior sample."""
    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import SVMPC, ExponentiatedUtility, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import PendulumModel

    g = golden("pend_k1")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    model = PendulumModel()
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=1.0, a_cov=4.0 * torch.eye(1), inst_cost_fn=inst_cost,
                      term_cost_fn=term_cost, params_sampling=None)
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), 4.0 * torch.eye(1))
    lik = ExponentiatedUtility(alpha=1.0, n_samples=S, controller=ctrl, model=model)
    sv = SVMPC(init_particles=torch.tensor(g["theta0"]), prior=prior, likelihood=lik, kernel=RBFKernel(), n_particles=N, n_steps=1,
               optimizer_class=torch.optim.SGD, lr=2.0, roll_strategy="resample")
    state = torch.tensor(g["state"][0, 0])
    sv.optimize(state, None, n_steps=1, eps=g["eps"][0][:1])
    th = sv.theta.clone()
    torch.manual_seed(9)
    last = sv.prior.sample([N])[..., -1, :]
    torch.manual_seed(9)
    sv.forward(state, None)
    ref = th.roll(-1, dims=-2)
    ref[..., -1, :] = last
    assert relerr(sv.theta.numpy(), ref.numpy()) < 1e-6
    a_seq, pw = sv.tick(state, None, n_steps=1)  # optimize + host draw + forward
    assert a_seq.shape == (H, 1) and abs(float(pw.sum()) - 1) < 5e-4


def test_gaussian_likelihood_sample_and_log_prob():
    """GaussianLikelihood.sample / log_prob (likelihoods.py:30-49): one-step predictions per parameter particle and the
    observation density - the object-protocol form of what the MPF kernel evaluates on the device."""
    from dust_amd.inference import GaussianLikelihood
    from dust_amd.models import PendulumModel

    model = PendulumModel(uncertain_params=("length", "mass"))
    lik = GaussianLikelihood(torch.tensor([3.0, 0.0]), obs_std=0.1, model=model, log_space=True)
    with pytest.raises(AssertionError):
        lik.sample(torch.zeros(4, 2))
    lik.condition(torch.tensor([1.3]), torch.tensor([2.9, -0.4]))
    theta = torch.log(torch.tensor([[0.9, 1.1], [1.2, 0.8], [1.0, 1.0]]))
    pred = lik.sample(theta)
    assert pred.shape == (3, 2)
    for i in range(3):
        m = PendulumModel(length=float(theta[i, 0].exp()), mass=float(theta[i, 1].exp()))
        ref = m.step(torch.tensor([[3.0, 0.0]]), torch.tensor([[1.3]]))
        assert relerr(pred[i].numpy(), ref[0].numpy()) < 1e-6
    lp = lik.log_prob(pred)
    d = pred - torch.tensor([2.9, -0.4])
    ref_lp = -0.5 * (d * d).sum(-1) / 0.01 - np.log(2 * np.pi * 0.01)
    assert lp.shape == (3, 1) and relerr(lp[:, 0].numpy(), ref_lp.numpy()) < 1e-5


def test_pendulum_driver_vs_reference_driver(golden):
    """the build's `run_pendulum_simulation` against the REFERENCE's own driver loop
    (dust/utils/simulations.py:13-190, run by tests/golden/make_golden_driver.py with a stand-in plant) over three control ticks of
    the dual-inference configuration: same call order per tick (optimize -> forward unless warming up -> plant step ->
    mpf.optimize), every random draw of the reference run replayed (MultiDISCO.draw_source).  Per-tick products: applied action,
    plant state, particle weights, the particles' first action after forward, the dynamics filter's particles."""
    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import MPF, GaussianLikelihood, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import PendulumModel
    from dust_amd.utils.simulations import run_pendulum_simulation

    g = golden("driver_pend_dual")
    N, H, S, M, Mp = (int(g[k]) for k in ("N", "H", "S", "M", "Mp"))
    steps, warm = int(g["steps"]), int(g["warm_up"])
    env_model = PendulumModel()
    init_state = torch.tensor(g["init_state"])
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), float(g["sigma"]) ** 2 * torch.eye(1))
    init_policies = torch.tensor(g["init_policies"])
    dyn_prior = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])), 1)
    ctrl = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H, action_samples=S,
                      params_samples=M, temperature=1.0, a_cov=float(g["sigma"]) ** 2 * torch.eye(1), inst_cost_fn=inst_cost,
                      term_cost_fn=term_cost, params_sampling=True, n_policies=N, params_log_space=False)
    lik = GaussianLikelihood(initial_obs=init_state, obs_std=float(g["obs_std"]), model=PendulumModel(uncertain_params=("length", "mass")),
                             log_space=False)
    mpf = MPF(init_particles=torch.tensor(g["mpf_init"]), likelihood=lik, optimizer_class=torch.optim.SGD, lr=float(g["mpf_lr"]),
              bw=float(g["mpf_bw"]), bw_scale=1.0)
    ctrl.draw_source = RecordedDraws(eps=list(g["eps"]), params=list(g["params"]))  # (deep-copied with the controller per episode)
    df = run_pendulum_simulation(
        init_state, init_policies, {"uncertain_params": ("length", "mass")}, dyn_prior,
        [{"length": float(g["true_length"]), "mass": float(g["true_mass"])}], ctrl, use_exact_model=False, use_svmpc=True,
        svmpc_kwargs=dict(init_particles=init_policies, prior=prior, kernel=RBFKernel(), n_particles=N, bw_scale=1.0, n_steps=1,
                          optimizer_class=torch.optim.SGD, lr=float(g["lr"])),
        lik_kwargs={"alpha": 1.0, "n_samples": S}, mpf=mpf, mpf_bw=float(g["mpf_bw"]), mpf_steps=int(g["mpf_steps"]), episodes=1,
        steps=steps, warm_up=warm)
    assert len(df) == steps
    for t in range(steps):
        # the state each tick started from (tick t+1's input = the plant state after tick t's action)
        if t + 1 < steps:
            got = np.array([df["Position"].iloc[t], df["Speed"].iloc[t]], np.float32)
            assert relerr(got, g["state_in"][t + 1]) < 1e-4, t
        dyn = np.asarray(df["DynParticles"].iloc[t], np.float32)
        assert relerr(dyn, g["mpf_x"][t]) < (1e-5 if t == 0 else 1e-3), t
        if t < warm:
            assert float(df["Actions"].iloc[t]) == 0.0
            continue
        k = t - warm
        ref_pw = g["p_weights"][k]
        pw = np.asarray(df["Weights"].iloc[t], np.float32)
        assert int(np.argmax(pw)) == int(np.argmax(ref_pw)), t
        assert relerr(pw, ref_pw) < 5e-3, t  # exp of O(1e3) log-weights (tests/test_gpu_parity.py)
        assert abs(float(df["Actions"].iloc[t]) - float(g["a_seq"][k][0, 0])) < 2e-3 * max(1.0, abs(float(g["a_seq"][k][0, 0]))), t
        assert relerr(np.asarray(df["PolParticles"].iloc[t], np.float32), g["theta_fwd"][k][:, 0, 0]) < 2e-3, t


@pytest.mark.parametrize("tag", ["run", "goal", "crash"])
def test_particle_episode_vs_reference_driver(golden, tag):
    """the build's `run_particle_episode` against the REFERENCE's own episode driver (dust/utils/simulations.py:197-260,
    the loop demo/particle_example.py:150-254 runs inline; fixtures by tests/golden/make_golden_episode.py): the mass-change event at
    steps // 4, zero actions while warming up, goal termination (|target - state| <= 1) and crash termination (cost inf), with every
    random draw of the reference run replayed.  Checked per tick: the plant state the tick started from, the particles after
    optimize, the chosen sequence and the particle weights; per episode: the number of steps run and the cumulative cost."""
    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import SVMPC, ExponentiatedUtility, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import Particle
    from dust_amd.utils.simulations import run_particle_episode

    g = golden("episode_part_" + tag)
    N, H, S, M = (int(g[k]) for k in ("N", "H", "S", "M"))
    sigma = float(g["sigma"])
    rec = dict(state_in=[], theta_opt=[], a_seq=[], p_weights=[])

    class RecSVMPC(SVMPC):
        def optimize(self, state, params_dist, *a, **k):
            rec["state_in"].append(np.asarray(state, np.float32).reshape(-1).copy())
            out = super().optimize(state, params_dist, *a, **k)
            rec["theta_opt"].append(self.theta.numpy().copy())
            return out

        def forward(self, state, params_dist, *a, **k):
            a_seq, pw = super().forward(state, params_dist, *a, **k)
            rec["a_seq"].append(a_seq.numpy().copy())
            rec["p_weights"].append(pw.numpy().copy())
            return a_seq, pw

    model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), sigma ** 2 * torch.eye(2))
    x = torch.tensor(g["dyn_means"])
    dyn = dist.MixtureSameFamily(dist.Categorical(torch.ones(x.shape[0])),
                                 dist.Independent(dist.MultivariateNormal(loc=x, covariance_matrix=float(g["dyn_bw"]) ** 2 * torch.eye(1)), 0))
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=1.0, a_cov=sigma ** 2 * torch.eye(2),
                      params_sampling=True, params_samples=M, params_log_space=True, inst_cost_fn=model.default_inst_cost,
                      term_cost_fn=model.default_term_cost)
    lik = ExponentiatedUtility(1.0, controller=ctrl, model=model, n_samples=S)
    sv = RecSVMPC(init_particles=torch.tensor(g["init_policies"]), prior=prior, likelihood=lik, kernel=RBFKernel(), n_particles=N,
                  bw_scale=1.0, n_steps=1, optimizer_class=torch.optim.SGD, lr=float(g["lr"]), weighted_prior=True)
    ctrl.draw_source = RecordedDraws(eps=list(g["eps"]), params=list(g["params"]))
    cum = run_particle_episode(torch.tensor(g["init_state"]), model, dyn, ctrl, use_svmpc=True, warm_up=int(g["warm_up"]), svmpc=sv,
                               load=float(g["load"]), steps=int(g["steps"]))
    n_run = int(g["steps_run"])
    assert len(rec["state_in"]) == n_run, (len(rec["state_in"]), n_run)  # same termination step
    ref_cum = float(g["cum_cost"])
    if np.isfinite(ref_cum):
        assert abs(float(cum) - ref_cum) < 2e-3 * abs(ref_cum), (float(cum), ref_cum)
    else:
        assert not np.isfinite(float(cum))  # crashed: cost inf (simulations.py:241)
    for t in range(n_run):
        # the plant trajectory: tick t starts where tick t - 1's action (and, from steps // 4 on, the heavier system) left the state
        assert elemerr(rec["state_in"][t], g["state_in"][t], floor=1.0) < 1e-4, t
        # (crash: every rollout starts inside the obstacle cost, 1e6 per step - the costs are O(1e7), one fp32 ulp of them is a
        # factor e on a softmax weight, and the particles after one step follow the weights)
        assert elemerr(rec["theta_opt"][t], g["theta_opt"][t]) < (5e-2 if tag == "crash" else 2e-3), t
    if "a_seq" in g:
        assert len(rec["a_seq"]) == len(g["a_seq"])
        for k in range(len(g["a_seq"])):
            ref_pw = g["p_weights"][k]
            assert int(np.argmax(rec["p_weights"][k])) == int(np.argmax(ref_pw)), k
            assert np.abs(rec["p_weights"][k] - ref_pw).max() < 5e-3, k  # exp of O(1e3) log-weights (tests/test_gpu_parity.py)
            assert elemerr(rec["a_seq"][k], g["a_seq"][k]) < 2e-3, k


def test_skid_steer_through_the_mirror_classes(golden):
    """`MultiDISCO(...).forward(state, SkidSteerRobot(...), ext_actions=...)` with `QuadraticCost` as the cost callables -
    the call the reference's controller accepts (disco.py:348-394) - against the reference's own result; an opaque callable raises."""
    import torch

    from dust_amd.controllers import MultiDISCO
    from dust_amd.costs import QuadraticCost
    from dust_amd.models import SkidSteerRobot

    g = golden("skid_nominal")
    N, H, S = int(g["N"]), int(g["H"]), int(g["S"])
    model = SkidSteerRobot(delta_t=float(g["dt"]))
    cost = QuadraticCost(g["goal"], g["w_state"], g["w_term"], g["w_ctrl"])
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=float(g["temperature"]), ctrl_penalty=1.0,
                      a_cov=float(g["sigma_a"]) ** 2 * torch.eye(2), inst_cost_fn=cost.inst_cost, term_cost_fn=cost.term_cost, params_sampling=None)
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    costs, states, actions, omega, _ = ctrl.forward(torch.tensor(g["state"]), model, None, ext_actions=torch.tensor(g["ext_actions"]))
    assert elemerr(costs.numpy(), g["costs"]) < 1e-5
    assert np.abs(states.numpy() - g["states"]).max() < 1e-5 * max(1.0, np.abs(g["states"]).max())
    assert relerr(ctrl.a_mat.numpy(), g["a_mat1"]) < 1e-4
    # the host-side cost object evaluates the same family (what a user would hand to the reference)
    x = torch.tensor(g["states"][0, :, :, :-1].reshape(-1, 5))
    a = torch.tensor(g["ext_actions"].reshape(-1, 2))
    tot = cost.inst_cost(x, a).reshape(S, N, H).sum(-1) + cost.term_cost(torch.tensor(g["states"][0, :, :, -1].reshape(-1, 5))).reshape(S, N)
    assert relerr(tot.numpy(), g["costs"]) < 1e-5
    bad = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=1.0, a_cov=0.09 * torch.eye(2),
                     inst_cost_fn=lambda s, a, **k: s.sum(-1), term_cost_fn=lambda s, **k: s.sum(-1), params_sampling=None)
    with pytest.raises(NotImplementedError):
        bad.forward(torch.tensor(g["state"]), model, None, ext_actions=torch.tensor(g["ext_actions"]))


@pytest.mark.parametrize("name", ["part_k1_noisy", "part_k1_velocity", "part_k2_noisy_vel"])
def test_particle_control_noise_and_velocity_through_the_class_api(golden, name):
    """`Particle(deterministic=False, noise_std=...)` - the reference's constructor default for `deterministic` - and
    `control_type="velocity"` through the reference's own classes (particle.py:145-153): the controller forwards both to the device
    rollouts, and the recorded draws of the reference run (policy noise, dynamics samples, control-channel noise) replayed through
    `MultiDISCO.draw_source` reproduce its costs (1e-5), rollout states (1e-5) and particles after every SVGD step (2e-3, whole chain)."""
    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import SVMPC, ExponentiatedUtility, get_gmm
    from dust_amd.kernels import RBF, RBFKernel, iid_mp
    from dust_amd.models import Particle

    g = golden(name)
    N, H, S, M = (int(g[k]) for k in ("N", "H", "S", "M"))
    env = dict(PARTICLE_ENV, deterministic=bool(int(g["deterministic"])), noise_std=torch.tensor(g["dyn_std"]), control_type=str(g["control_type"]))
    if env["control_type"] == "velocity":
        env.update(init_state=[-4.0, -3.0], target_state=[4.0, 4.5])
    model = Particle(**env, uncertain_params=["mass"], mass=torch.tensor(2.0))
    assert model.observation_space.dim == int(g["ds"])
    scalar = bool(int(g["params_scalar_event"]))
    if scalar:
        pdist = dist.Normal(2.0, 0.1)
    else:
        pdist = dist.MixtureSameFamily(dist.Categorical(torch.ones(16)),
                                       dist.Independent(dist.MultivariateNormal(torch.zeros(16, 1), 0.25 * torch.eye(1)), 0))
    sig = float(g["sigma_a"])
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=float(g["temperature"]), a_cov=sig ** 2 * torch.eye(2),
                      params_sampling=True, params_samples=M, params_log_space=bool(int(g["params_log_space"])),
                      inst_cost_fn=model.default_inst_cost, term_cost_fn=model.default_term_cost)
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    kind = str(g["kernel_kind"])
    kernel = RBFKernel() if kind == "K1" else iid_mp(base_kernel=RBF(bandwidth=-1), ctrl_dim=2, indep_controls=True)
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), float(g["sigma_p"]) ** 2 * torch.eye(2))
    lik = ExponentiatedUtility(float(g["alpha"]), controller=ctrl, model=model, n_samples=S)
    sv = SVMPC(init_particles=torch.tensor(g["theta0"]), prior=prior, likelihood=lik, kernel=kernel, n_particles=N, bw_scale=1.0, n_steps=1,
               optimizer_class=torch.optim.SGD, lr=float(g["lr"]), weighted_prior=bool(int(g["weighted_prior"])))
    T, K = g["eps"].shape[:2]
    cz = list(g["ctrl_noise"].reshape((T * K,) + g["ctrl_noise"].shape[2:])) if "ctrl_noise" in g else None
    # (a) one likelihood sample + one MultiDISCO.forward from the reference's first inputs: costs and states at 1e-5
    ctrl.draw_source = RecordedDraws(params=[g["params"][0, 0]] * 2, ctrl_noise=None if cz is None else [cz[0]] * 2)
    state0 = torch.tensor(g["state"][0, 0])
    costs, actions = lik.sample(torch.tensor(g["theta0"]), state0, pdist, eps=g["eps"][0, 0])
    assert np.array_equal(actions.numpy(), g["actions"][0, 0])
    assert elemerr(costs.numpy(), g["costs"][0, 0]) < 1e-5
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    _, states, _, _, _ = ctrl.forward(state0, model, pdist, ext_actions=torch.tensor(g["actions"][0, 0]))
    assert elemerr(states.numpy(), g["states_iter0"][0]) < 1e-5
    # (b) the optimisation chain
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    ctrl.draw_source = RecordedDraws(eps=list(g["eps"].reshape((T * K,) + g["eps"].shape[2:])),
                                     params=list(g["params"].reshape((T * K,) + g["params"].shape[2:])), ctrl_noise=cz)
    for t in range(T):
        state = torch.tensor(g["state"][t, 0])
        sv.optimize(state, pdist, n_steps=K)
        scale = np.abs(g["theta_after"][t, K - 1]).max()
        assert np.abs(sv.theta.numpy() - g["theta_after"][t, K - 1]).max() / scale < 2e-3, (name, t)
        sv.forward(state, pdist)
        sv.theta = torch.tensor(g["tick_theta_rolled"][t])  # re-sync with the reference (stage-local checks)
    assert ctrl._ctx.tick_stats()["tick2"] == 0  # (these configurations run on the launch-per-iteration path)


def test_mpf_control_noise_through_the_class_api(golden):
    """MPF over `Particle(deterministic=False)` through the reference's classes; recorded draws via `MPF.draw_source`."""
    from dust_amd.inference import MPF, GaussianLikelihood
    from dust_amd.models import Particle

    g = golden("mpf_part_noisy")
    env = dict(PARTICLE_ENV, deterministic=False, noise_std=torch.tensor(g["dyn_std"]))
    model = Particle(**env, uncertain_params=["mass"], mass=torch.tensor(2.0))
    lik = GaussianLikelihood(initial_obs=torch.tensor(g["obs0"]), obs_std=float(g["obs_std"]), model=model, log_space=True)
    mpf = MPF(init_particles=torch.tensor(g["x0"]), likelihood=lik, optimizer_class=torch.optim.SGD, lr=float(g["lr"]), bw=float(g["bw"]))
    mpf.draw_source = RecordedDraws(mpf_noise=list(g["noise1"]) + list(g["noise2"]))
    n = int(g["n_steps"])
    grads, _ = mpf.optimize(torch.tensor(g["action"]), torch.tensor(g["obs1"]), bw=float(g["bw"]), n_steps=n)
    assert elemerr(mpf.x.numpy(), g["x_final"]) < 1e-5 and relerr(grads.numpy(), g["grad_norms"]) < 2e-4
    mpf.optimize(torch.tensor(g["action2"]), torch.tensor(g["obs2"]), bw=float(g["bw"]), n_steps=n)
    assert elemerr(mpf.x.numpy(), g["x_final2"]) < 1e-5


@pytest.mark.parametrize("fixture", ["driver_pend_dual", "driver_pend_dual_bwnull"])
def test_dual_svmpc_facade_vs_reference_driver(golden, fixture):
    """(`driver_pend_dual_bwnull`: demo/pendulum_config.yaml's own `mpf_bandwidth: null` - the filter's first prior from bw_silverman
    of its particles, every filter update with bw = silvermans_rule of the pooled particles, mpf.py:31-36, 68-73 - the bandwidths
    the reference used are part of the fixture.)
    BASELINE north_star's named surface, `DualSVMPC` with step() / forward() (the reference composes the two inferences by hand:
    dust/utils/simulations.py:104-138), held to the golden of the reference's OWN dual loop (`driver_pend_dual`): forward(state) =
    optimize + forward (zero plan while warming up), step(action, new_state) = the filter update; tick() = forward, plant, step.
    A deep copy continues identically (the loop deep-copies controller and filter per episode, simulations.py:62,78)."""
    import copy

    from dust_amd.controllers import DualSVMPC, MultiDISCO
    from dust_amd.inference import MPF, SVMPC, ExponentiatedUtility, GaussianLikelihood, get_gmm
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import PendulumModel

    g = golden(fixture)
    bw_arg = None if float(g["mpf_bw"]) < 0 else float(g["mpf_bw"])
    N, H, S, M = (int(g[k]) for k in ("N", "H", "S", "M"))
    steps, warm = int(g["steps"]), int(g["warm_up"])
    env_model = PendulumModel()
    init_state = torch.tensor(g["init_state"])
    init_policies = torch.tensor(g["init_policies"])
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), float(g["sigma"]) ** 2 * torch.eye(1))
    ctrl = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H, action_samples=S,
                      params_samples=M, temperature=1.0, a_cov=float(g["sigma"]) ** 2 * torch.eye(1), inst_cost_fn=inst_cost,
                      term_cost_fn=term_cost, params_sampling=True, n_policies=N, params_log_space=False)
    ctrl.a_mat = init_policies.clone()
    ctrl.return_rollouts = False
    ctrl.draw_source = RecordedDraws(eps=list(g["eps"]), params=list(g["params"]))
    mpf_model = PendulumModel(uncertain_params=("length", "mass"))
    mpf = MPF(init_particles=torch.tensor(g["mpf_init"]), likelihood=GaussianLikelihood(initial_obs=init_state, obs_std=float(g["obs_std"]),
                                                                                         model=mpf_model, log_space=False),
              optimizer_class=torch.optim.SGD, lr=float(g["mpf_lr"]), bw=bw_arg, bw_scale=1.0)
    # the model the controller rolls out: the filter prior's mean parameters (use_exact_model=False, simulations.py:45-47)
    model = PendulumModel(length=mpf.prior.mean[0], mass=mpf.prior.mean[1], uncertain_params=("length", "mass"))
    sv = SVMPC(likelihood=ExponentiatedUtility(alpha=1.0, n_samples=S, controller=ctrl, model=model), init_particles=init_policies, prior=prior,
               kernel=RBFKernel(), n_particles=N, bw_scale=1.0, n_steps=1, optimizer_class=torch.optim.SGD, lr=float(g["lr"]))
    dual = DualSVMPC(sv, mpf, mpf_bw=bw_arg, mpf_steps=int(g["mpf_steps"]), warm_up=warm)
    assert dual.dyn_dist is mpf.prior and dual.controller is ctrl
    plant_model = PendulumModel(g=10.0, length=float(g["true_length"]), mass=float(g["true_mass"]))  # the driver's gym stand-in

    def plant(state, action):
        return plant_model.step(state, torch.as_tensor(action, dtype=torch.float).clamp(-2.0, 2.0).reshape(1, -1)).reshape(1, -1)

    state = init_state.reshape(1, -1)
    twin = None
    for t in range(steps):
        if t == steps - 1:
            twin = copy.deepcopy(dual)  # (taken before the last tick: it must produce the same last tick)
            twin_state = state.clone()
        assert relerr(state.numpy().reshape(-1), g["state_in"][t]) < 1e-4, t
        action, state, pw = dual.tick(state, plant)
        if "mpf_bw_used" in g:  # the bandwidth each filter update ran with (Silverman's rule of the particles when none is given)
            assert abs(float(dual.last_bw) - float(g["mpf_bw_used"][t])) < (1e-6 if bw_arg is not None else 1e-5 * (1 + 50 * t)) * float(g["mpf_bw_used"][t]) + 1e-9, t
        assert relerr(dual.dyn_particles.numpy(), g["mpf_x"][t]) < (1e-5 if t == 0 else 1e-3), t
        if t < warm:
            assert pw is None and float(action.abs().max()) == 0.0
            continue
        k = t - warm
        assert int(pw.argmax()) == int(np.argmax(g["p_weights"][k])) and relerr(pw.numpy(), g["p_weights"][k]) < 5e-3, t
        assert abs(float(action[0]) - float(g["a_seq"][k][0, 0])) < 2e-3 * max(1.0, abs(float(g["a_seq"][k][0, 0]))), t
        assert relerr(dual.theta.numpy()[:, 0, 0], g["theta_fwd"][k][:, 0, 0]) < 2e-3, t
    # the deep copy: same recorded draws (copied with the controller), same tick
    a2, s2, pw2 = twin.tick(twin_state, plant)
    assert torch.equal(a2, action) and torch.equal(s2, state) and torch.equal(pw2, pw)
    assert twin.controller is not ctrl and twin.mpf is not mpf and twin.dyn_dist is twin.mpf.prior
    # the control half alone (no filter): forward() works, step() is a no-op
    solo = DualSVMPC(sv, None, dyn_dist=mpf.prior, warm_up=0)
    ctrl.draw_source = None
    a_seq, pw = solo.forward(state)
    assert a_seq.shape == (H, 1) and abs(float(pw.sum()) - 1.0) < 1e-3 and solo.step(a_seq[0], state) == (None, None)


@pytest.mark.parametrize("name", ["part_k1_fullcov", "part_k2_fullcov"])
def test_full_covariances_through_the_class_api(golden, name):
    """VERDICT r4 missing item 4: `MultiDISCO(a_cov=<full SPD matrix>)` (disco.py:91-98: policy noise through cholesky(a_cov), control cost
    through inverse(a_cov)) and a prior GMM with a full component covariance (get_gmm svgd.py:84-89; likelihoods.py:85-87) - both were
    rejected until round 5.  The reference's own run (tests/golden/make_golden_r5.py run_svmpc_cov), its draws replayed through
    MultiDISCO.draw_source: costs of one likelihood sample at 1e-5, the particles after every SVGD step and forward's weights."""
    import torch.distributions as dist

    from dust_amd.controllers import MultiDISCO
    from dust_amd.inference import SVMPC, ExponentiatedUtility, get_gmm
    from dust_amd.kernels import RBF, RBFKernel, iid_mp
    from dust_amd.models import Particle

    g = golden(name)
    N, H, S, M = (int(g[k]) for k in ("N", "H", "S", "M"))
    model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    pdist = dist.MixtureSameFamily(dist.Categorical(torch.ones(16)), dist.Independent(dist.MultivariateNormal(torch.zeros(16, 1), 0.25 * torch.eye(1)), 0))
    a_reg, temp = float(g["a_reg"]), float(g["temperature"])
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=temp, ctrl_penalty=1.0 - a_reg / temp, a_cov=torch.tensor(g["a_cov"]),
                      params_sampling=True, params_samples=M, params_log_space=True, inst_cost_fn=model.default_inst_cost,
                      term_cost_fn=model.default_term_cost)
    assert torch.allclose(ctrl.a_pre, torch.inverse(torch.tensor(g["a_cov"])))
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    kernel = RBFKernel() if str(g["kernel_kind"]) == "K1" else iid_mp(base_kernel=RBF(bandwidth=-1), ctrl_dim=2, indep_controls=True)
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), torch.tensor(g["p_cov"]))
    lik = ExponentiatedUtility(float(g["alpha"]), controller=ctrl, model=model, n_samples=S)
    sv = SVMPC(init_particles=torch.tensor(g["theta0"]), prior=prior, likelihood=lik, kernel=kernel, n_particles=N, bw_scale=1.0, n_steps=1,
               optimizer_class=torch.optim.SGD, lr=float(g["lr"]), weighted_prior=True)
    T, K = g["eps"].shape[:2]
    ctrl.draw_source = RecordedDraws(params=[g["params"][0, 0]])
    costs, actions = lik.sample(torch.tensor(g["theta0"]), torch.tensor(g["state"][0, 0]), pdist, eps=g["eps"][0, 0])
    assert elemerr(actions.numpy(), g["actions"][0, 0]) < 1e-6 and elemerr(costs.numpy(), g["costs"][0, 0]) < 1e-5
    ctrl.a_mat = torch.tensor(g["a_mat0"])
    ctrl.draw_source = RecordedDraws(eps=list(g["eps"].reshape((T * K,) + g["eps"].shape[2:])), params=list(g["params"].reshape((T * K,) + g["params"].shape[2:])))
    for t in range(T):
        state = torch.tensor(g["state"][t, 0])
        sv.optimize(state, pdist, n_steps=K)
        scale = np.abs(g["theta_after"][t, K - 1]).max()
        assert np.abs(sv.theta.numpy() - g["theta_after"][t, K - 1]).max() / scale < 2e-3, (name, t)
        a_seq, pw = sv.forward(state, pdist)
        assert int(pw.argmax()) == int(np.argmax(g["tick_p_weights"][t]))
        sv.theta = torch.tensor(g["tick_theta_rolled"][t])


@pytest.mark.parametrize("Mp,P,kind", [(256, 2, "normal"), (10, 2, "normal"), (64, 1, "ties"), (33, 2, "tight_iqr"), (1024, 2, "normal"), (2, 1, "normal")])
def test_device_silverman_rule_matches_the_host_rule(Mp, P, kind):
    """dust_mpf_silverman: MPF.optimize's default bandwidth (mpf.py:68-73, `silvermans_rule(self.x.view(-1, 1)) * bw_scale`) evaluated on
    the device - float64, numpy's linear-interpolation quartiles - against the host restatement of KDEpy's rule
    (dust_amd/inference/mpf.py; third party, parity unpinned): equal to fp32 rounding, for ties, a zero IQR and two values as well."""
    from dust_amd import MpfContext
    from dust_amd.inference.mpf import silvermans_rule

    rng = np.random.default_rng(Mp + P)
    x = (1.0 + 0.2 * rng.standard_normal((Mp, P))).astype(np.float32)
    if kind == "ties":
        x = np.round(x * 8) / 8
    elif kind == "tight_iqr":  # more than half of the values identical: IQR = 0, the rule falls back to the standard deviation
        x[: (3 * Mp) // 4] = 1.0
    state = np.array([3.0, 0.0], np.float32)
    for scale in (1.0, 0.5):
        m = MpfContext(x, state, model="pendulum", uncertain_params=("length", "mass")[:P], obs_std=0.1, lr=1e-3, init_bw=0.1, bw_scale=scale)
        want = silvermans_rule(x.astype(np.float64).reshape(-1, 1)) * scale
        got = m.silverman()
        assert abs(got - np.float32(want)) <= 2e-7 * abs(want), (got, want)
        m.close()


def test_fused_dual_tick_equals_its_pieces(monkeypatch):
    """(Both sides on the launch-per-iteration kernels - DUST_NO_TICK2: the dual tick's staged samples keep the controller off the
    one-launch tick, whose sums run in another order.)
    dust_dual_tick (DualSVMPC(fused=True)): the filter update with Silverman's bandwidth on the device, the controller's dynamics samples
    drawn from the refreshed filter prior on the device, the control tick - ONE call - against the same pieces called one by one through the
    C ABI with the same Philox key (dust_mpf_silverman, dust_mpf_optimize, dust_mpf_prior_sample -> host -> dust_svmpc_tick): bit-identical
    action sequences, weights, particles and filter particles over five control periods."""
    from dust_amd import Context, MpfContext

    monkeypatch.setenv("DUST_NO_TICK2", "1")
    N, S, M, H, K, Mp = 64, 32, 4, 12, 2, 48
    rng = np.random.default_rng(9)
    mu = rng.standard_normal((N, H, 1)).astype(np.float32)
    th = (mu + rng.standard_normal((N, H, 1))).astype(np.float32)
    x0 = rng.uniform(0.6, 1.3, (Mp, 2)).astype(np.float32)

    def make():
        c = Context(model="pendulum", N=N, S=S, M=M, H=H, kernel="K1", lr=0.5, sigma_a=2.0, sigma_p=2.0, uncertain_params=("length", "mass"), seed=5)
        c.set_theta(th); c.set_prior(mu); c.set_a_mat(th)
        m = MpfContext(x0, np.array([3.0, 0.0], np.float32), model="pendulum", uncertain_params=("length", "mass"), obs_std=0.1, lr=1e-3, init_bw=0.1)
        return c, m

    def plant(st, a):
        thd = np.float32(np.clip(st[1] + 0.05 * (14.7 * np.sin(st[0]) + 3.0 * np.clip(a, -2, 2)), -8, 8))
        return np.array([st[0] + thd * 0.05, thd], np.float32)

    ca, ma = make()
    cb, mb = make()
    sa = sb = np.array([3.0, 0.0], np.float32)
    prev = None
    for t in range(5):
        a1, p1, bw1 = ca.dual_tick(ma, sa, prev, K, mpf_steps=6, mpf_bw=None, seed=100 + t)
        if prev is not None:
            bw2 = mb.silverman()
            mb.optimize(prev, sb, bw2, 6)
            assert bw1 == bw2
        params = mb.prior_sample(K * M, 100 + t).reshape(K, M, 2)
        a2, p2 = cb.svmpc_tick(sb, K, None, params)
        assert np.array_equal(a1, a2) and np.array_equal(p1, p2), t
        assert np.array_equal(ma.get_particles(), mb.get_particles()), t
        prev = a1[0].copy()
        sa = sb = plant(sa, float(a1[0, 0]))
    assert np.array_equal(ca.get_theta(), cb.get_theta())
    for o in (ca, cb, ma, mb):
        o.close()


def test_dual_svmpc_fused_runs_the_reference_loop_shape(golden):
    """DualSVMPC(fused=True) on the shapes of the reference's dual driver (`driver_pend_dual_bwnull`: mpf_bandwidth null), own draws: the
    filter update noted by step() is carried out by the next forward(); bandwidths are Silverman's rule of the particles that update
    started from; every output finite and normalised; reading the filter between step() and forward() carries the update out first."""
    from dust_amd.controllers import DualSVMPC, MultiDISCO
    from dust_amd.inference import MPF, SVMPC, ExponentiatedUtility, GaussianLikelihood, get_gmm
    from dust_amd.inference.mpf import silvermans_rule
    from dust_amd.kernels import RBFKernel
    from dust_amd.models import PendulumModel

    g = golden("driver_pend_dual_bwnull")
    N, H, S, M = (int(g[k]) for k in ("N", "H", "S", "M"))
    env_model = PendulumModel()
    init_state = torch.tensor(g["init_state"])
    init_policies = torch.tensor(g["init_policies"])
    prior = get_gmm(torch.tensor(g["mu0"]), torch.ones(N), float(g["sigma"]) ** 2 * torch.eye(1))
    ctrl = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H, action_samples=S,
                      params_samples=M, temperature=1.0, a_cov=float(g["sigma"]) ** 2 * torch.eye(1), inst_cost_fn=inst_cost,
                      term_cost_fn=term_cost, params_sampling=True, n_policies=N, params_log_space=False)
    ctrl.a_mat = init_policies.clone()
    ctrl.return_rollouts = False
    mpf = MPF(init_particles=torch.tensor(g["mpf_init"]), likelihood=GaussianLikelihood(initial_obs=init_state, obs_std=float(g["obs_std"]),
                                                                                         model=PendulumModel(uncertain_params=("length", "mass")), log_space=False),
              optimizer_class=torch.optim.SGD, lr=float(g["mpf_lr"]), bw=None, bw_scale=1.0)
    model = PendulumModel(length=mpf.prior.mean[0], mass=mpf.prior.mean[1], uncertain_params=("length", "mass"))
    sv = SVMPC(likelihood=ExponentiatedUtility(alpha=1.0, n_samples=S, controller=ctrl, model=model), init_particles=init_policies, prior=prior,
               kernel=RBFKernel(), n_particles=N, bw_scale=1.0, n_steps=1, optimizer_class=torch.optim.SGD, lr=float(g["lr"]))
    dual = DualSVMPC(sv, mpf, mpf_bw=None, mpf_steps=int(g["mpf_steps"]), warm_up=1, fused=True, seed=3)
    plant_model = PendulumModel(g=10.0, length=float(g["true_length"]), mass=float(g["true_mass"]))

    def plant(state, action):
        return plant_model.step(state, torch.as_tensor(action, dtype=torch.float).clamp(-2.0, 2.0).reshape(1, -1)).reshape(1, -1)

    state = init_state.reshape(1, -1)
    for t in range(6):
        x_before = mpf.x.numpy().copy() if t >= 1 else None  # (the filter is current here: the previous forward() ran its update)
        action, state, pw = dual.tick(state, plant)
        if t == 0:
            assert pw is None and float(action.abs().max()) == 0.0  # warming up: zero action, unfused
            continue
        assert torch.isfinite(action).all() and abs(float(pw.sum()) - 1.0) < 1e-4, t
        if t >= 2:  # this forward() carried out the update step() noted at t - 1, with Silverman's rule of the particles it started from
            assert dual.last_bw is not None and abs(dual.last_bw - silvermans_rule(x_before.reshape(-1, 1))) < 1e-6 * dual.last_bw, t
    assert dual._pending is not None
    x_now = dual.dyn_particles  # reading the filter carries the noted update out
    assert dual._pending is None and torch.isfinite(x_now).all()
