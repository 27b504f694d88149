#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE (lubaroli/dust) in this container.

TEST INFRASTRUCTURE.  Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (build container only); writes tests/golden/*.npz (small, committed).

The reference has no tests and no golden vectors of its own, so every fixture here is
produced by importing the reference through `oracle/ref_shim.py` and calling its own classes:
  dust.controllers.disco.MultiDISCO, dust.inference.svmpc.SVMPC, dust.inference.mpf.MPF,
  dust.inference.likelihoods.*, dust.inference.svgd.get_gmm, dust.kernels.*, dust.models.{pendulum,particle}.
All random draws the reference makes (policy noise via MultivariateNormal.rsample -> _standard_normal, and the
dynamics-parameter samples) are RECORDED and stored with the outputs, so the build can be driven with identical
noise (SURVEY.md section 7 "RNG parity").

What is reference-produced vs derived:
  * everything named in OUTPUT_KEYS below is a value the reference computed itself;
  * `grad_pri` is torch.autograd.grad of the reference's own `prior.log_prob(x).sum()` (the call at svmpc.py:41);
  * nothing here re-implements the reference's math.

Third-party caveat (parity unpinned, see oracle/ref_shim.py): kernel mode K1 uses our gpytorch-RBFKernel stand-in,
and MPF's `bw=None` path would use our KDEpy stand-in, so MPF fixtures pass an explicit bw.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

ref_shim.install()

import torch  # noqa: E402
import torch.distributions as dist  # noqa: E402
import torch.distributions.multivariate_normal as mvn_mod  # noqa: E402
from dust.controllers.disco import MultiDISCO  # noqa: E402
from dust.inference.likelihoods import (  # noqa: E402
    ExpectedCost,
    ExponentiatedUtility,
    GaussianLikelihood,
)
from dust.inference.mpf import MPF  # noqa: E402
from dust.inference.svgd import get_gmm  # noqa: E402
from dust.inference.svmpc import SVMPC  # noqa: E402
from dust.kernels.base_kernels import RBF  # noqa: E402
from dust.kernels.composite_kernels import iid_mp  # noqa: E402
from dust.models.particle import Particle  # noqa: E402
from dust.models.pendulum import PendulumModel  # noqa: E402
from dust.utils.obstacle_map import generate_obstacle_map, get_obst_preset  # noqa: E402

torch.autograd.set_detect_anomaly(False)  # only a NaN detector (svmpc.py:11); no numeric effect
torch.set_num_threads(4)

OUT = os.path.dirname(os.path.abspath(__file__))

# ---------------------------------------------------------------- RNG recording
_REC = []
_orig_std_normal = mvn_mod._standard_normal


def _rec_std_normal(shape, dtype, device):
    out = _orig_std_normal(shape, dtype, device)
    _REC.append(out.detach().clone())
    return out


mvn_mod._standard_normal = _rec_std_normal


class RecordingDist:
    """Wraps a torch distribution, recording `.sample()` draws (disco.py:168-172 is the only consumer)."""

    def __init__(self, d):
        self.d = d
        self.draws = []

    @property
    def event_shape(self):
        return self.d.event_shape

    @property
    def mean(self):
        return self.d.mean

    def sample(self, shape):
        s = self.d.sample(shape)
        self.draws.append(s.detach().clone())
        return s

    def log_prob(self, x):
        return self.d.log_prob(x)


# ---------------------------------------------------------------- cost functions of demo/pendulum_example.py:21-28
def pend_inst_cost(states, controls=None, n_pol=1, debug=None):
    th, thd = states.chunk(2, dim=1)
    return 50.0 * (th.cos() - 1) ** 2 + 1.0 * thd ** 2


def pend_term_cost(states, n_pol=1, debug=None):
    return pend_inst_cost(states).squeeze()


PARTICLE_ENV = dict(  # demo/particle_config.yaml:39-61
    dt=0.015,
    control_type="acceleration",
    noise_std=[0.1, 0.1],
    init_state=[-9.0, -9.0, 0, 0],
    target_state=[9.0, 9.0, 0, 0],
    can_crash=True,
    with_obstacle=True,
    deterministic=True,
    cost_params=dict(w_qpos=0.5, w_qvel=0.25, w_ctrl=0.2, w_obs=1.0e6, w_qpos_T=1.0e3, w_qvel_T=0.1),
    obst_preset="grid_4x4",
    obst_width=2.1,
    max_speed=5,
    max_accel=10,
    map_cell_size=0.1,
    map_size=[22, 22],
    map_type="direct",
)


def npf(t):
    return t.detach().cpu().numpy().copy()


# ---------------------------------------------------------------- SVMPC scenarios
def run_svmpc(
    tag,
    model_kind,
    N,
    H,
    S,
    M,
    kernel_kind="K1",
    k2_bandwidth=-1,
    lik_kind="ExponentiatedUtility",
    weighted_prior=False,
    roll_strategy="repeat",
    n_iters=2,
    n_ticks=2,
    alpha=1.0,
    seed=0,
    params_kind=None,
    state0=None,
    theta_shrink=1.0,
    ctrl_penalty=1.0,
    optimizer="SGD",
    lr_override=None,
    k1_f64=False,
):
    torch.manual_seed(seed)
    params_log_space = False
    pdist = None
    if model_kind == "pendulum":
        da, sigma_a, sigma_p, lr = 1, 2.0, 2.0, 2.0  # pendulum_config.yaml:15-20
        uncertain = ("length", "mass") if params_kind else None
        model = PendulumModel(uncertain_params=uncertain)
        inst_fn, term_fn = pend_inst_cost, pend_term_cost
        state = torch.tensor([3.0, 0.0] if state0 is None else state0)
        if params_kind == "uniform2":  # pendulum_example.py:81-83
            pdist = RecordingDist(dist.Independent(dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])), 1))
    else:
        da, sigma_a, sigma_p, lr = 2, 5.0, 5.0, 100.0  # particle_config.yaml:12-23
        model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
        inst_fn, term_fn = model.default_inst_cost, model.default_term_cost
        state = torch.tensor([-9.0, -9.0, 0.0, 0.0] if state0 is None else state0)
        if params_kind == "logmass_gmm":  # an MPF prior: GMM over log-mass particles, bw 0.5 (particle_config.yaml:35)
            x = dist.Normal(2.0, 0.1).sample([16, 1]).clamp(min=1e-6).log()
            mix = dist.Categorical(torch.ones(16))
            comp = dist.Independent(dist.MultivariateNormal(loc=x, covariance_matrix=0.5 ** 2 * torch.eye(1)), 0)
            pdist = RecordingDist(dist.MixtureSameFamily(mix, comp))
            params_log_space = True
        elif params_kind == "scalar_normal":  # particle_example.py:55 when use_mpf is false (scalar event)
            pdist = RecordingDist(dist.Normal(2.0, 0.1))

    if lr_override is not None:
        lr = lr_override
    mu0 = torch.randn(N, H, da)
    prior = get_gmm(mu0, torch.ones(N), sigma_p ** 2 * torch.eye(da))
    theta0 = prior.sample([N])
    if theta_shrink != 1.0:
        # pull particles together so the RBF Gram matrix is not numerically the identity
        theta0 = theta0.mean(0, keepdim=True) + theta_shrink * (theta0 - theta0.mean(0, keepdim=True))
        mu0 = theta0 + 0.05 * torch.randn_like(theta0)
        prior = get_gmm(mu0, torch.ones(N), sigma_p ** 2 * torch.eye(da))

    controller = MultiDISCO(
        model.observation_space,
        model.action_space,
        H,
        N,
        S,
        temperature=1.0 / alpha,
        ctrl_penalty=ctrl_penalty,
        a_cov=sigma_a ** 2 * torch.eye(da),
        inst_cost_fn=inst_fn,
        term_cost_fn=term_fn,
        params_sampling=(pdist is not None),
        params_samples=M,
        params_log_space=params_log_space,
    )
    controller.a_mat = theta0.detach().clone()  # simulations.py:63
    if kernel_kind == "K1":
        kernel = ref_shim.RBFKernel()
    elif kernel_kind == "K2":
        kernel = iid_mp(base_kernel=RBF(bandwidth=k2_bandwidth), ctrl_dim=da, indep_controls=True)
    elif kernel_kind == "K2shared":
        kernel = iid_mp(base_kernel=RBF(bandwidth=k2_bandwidth), ctrl_dim=da, indep_controls=False)
    else:
        raise ValueError(kernel_kind)
    lik_cls = ExponentiatedUtility if lik_kind == "ExponentiatedUtility" else ExpectedCost
    lik = lik_cls(alpha=alpha, n_samples=S, controller=controller, model=model)
    svmpc = SVMPC(
        init_particles=theta0.detach().clone(),
        prior=prior,
        likelihood=lik,
        kernel=kernel,
        n_particles=N,
        bw_scale=1.0,
        n_steps=1,
        optimizer_class=torch.optim.SGD if optimizer == "SGD" else torch.optim.Adam,
        lr=lr,
        weighted_prior=weighted_prior,
        roll_strategy=roll_strategy,
    )

    g = dict(
        N=N, H=H, S=S, M=controller.n_params, da=da, ds=model.observation_space.dim,
        sigma_a=sigma_a, sigma_p=sigma_p, alpha=alpha, lr=lr, temperature=1.0 / alpha, a_reg=controller.a_reg,
        n_iters=n_iters, n_ticks=n_ticks, weighted_prior=int(weighted_prior), params_log_space=int(params_log_space),
        params_scalar_event=int(params_kind == "scalar_normal"),
        theta0=npf(theta0), mu0=npf(mu0), mix0=np.ones(N, np.float32), a_mat0=npf(controller.a_mat),
    )
    keys = ["state", "eps", "params", "params_log_p", "actions", "costs", "omega_amat", "a_mix", "grad_pri", "phi",
            "theta_after", "states_iter0", "theta_in", "score", "phi_f64"]
    # K1 third-party boundary (k1_f64): record the score the reference hands to its kernel branch (the tensordot operand at
    # svmpc.py:83) and evaluate that same branch - svmpc.py:76-83, the gpytorch-semantics kernel - in float64 on the same inputs
    _orig_tensordot = torch.tensordot
    _td = []

    def _rec_tensordot(a_, b_, dims=2, **kw):
        _td.append((a_.detach().clone(), b_.detach().clone()))
        return _orig_tensordot(a_, b_, dims, **kw)
    per = {k: [] for k in keys}
    tick = {k: [] for k in ["log_l", "log_p", "p_weights", "a_seq", "theta_rolled", "prior_means", "prior_probs"]}

    for t in range(n_ticks):
        for k in range(n_iters):
            per["state"].append(npf(state))
            # prior score exactly as svmpc.py:38-41 obtains it
            x = svmpc.theta.detach().clone().requires_grad_(True)
            per["grad_pri"].append(npf(torch.autograd.grad(svmpc.prior.log_prob(x).sum(), x)[0]))
            _REC.clear()
            if pdist is not None:
                pdist.draws.clear()
            if k1_f64:
                per["theta_in"].append(npf(svmpc.theta))
                _td.clear()
                torch.tensordot = _rec_tensordot
            svmpc.optimize(state, pdist, n_steps=1)  # svmpc.py:97 -> step -> phi -> likelihood.sample -> forward
            if k1_f64:
                torch.tensordot = _orig_tensordot
                k_xx32, score32 = _td[-1]  # svmpc.py:83
                assert tuple(score32.shape) == (N, H, da) and tuple(k_xx32.shape) == (N, N)
                per["score"].append(npf(score32))
                x64 = torch.from_numpy(per["theta_in"][-1]).double().requires_grad_(True)
                k64 = ref_shim.RBFKernel()
                k64.raw_lengthscale = k64.raw_lengthscale.double()
                kxx = k64(x64.flatten(1, -1), x64.detach().clone().flatten(1, -1)).evaluate()
                grad_k = torch.autograd.grad(kxx.sum(), x64)[0]
                per["phi_f64"].append(npf(grad_k + _orig_tensordot(kxx.detach(), score32.double(), 1) / x64.size(0)))
            eps = [r for r in _REC if tuple(r.shape) == (S, N, H, da)]
            assert len(eps) == 1, [tuple(r.shape) for r in _REC]
            per["eps"].append(npf(eps[0]))
            if pdist is not None:
                assert len(pdist.draws) == 1
                p = pdist.draws[0]
                per["params"].append(npf(p.reshape(controller.n_params, -1)))
                per["params_log_p"].append(npf(lik.params_log_p))
            per["actions"].append(npf(lik.last_actions))
            per["costs"].append(npf(lik.last_costs))
            per["omega_amat"].append(npf(controller.a_mat))
            per["a_mix"].append(npf(controller.a_mix))
            per["phi"].append(npf(-svmpc.theta.grad))
            per["theta_after"].append(npf(svmpc.theta))
            if k == 0:
                per["states_iter0"].append(npf(lik.last_states))
        with torch.no_grad():
            tick["log_l"].append(npf(lik.log_prob(lik.last_costs)))
            tick["log_p"].append(npf(svmpc.prior.log_prob(svmpc.theta)))
        a_seq, p_w = svmpc.forward(state, pdist)  # svmpc.py:172
        tick["p_weights"].append(npf(p_w))
        tick["a_seq"].append(npf(a_seq))
        tick["theta_rolled"].append(npf(svmpc.theta))
        tick["prior_means"].append(npf(svmpc.prior.component_distribution.base_dist.loc))
        tick["prior_probs"].append(npf(svmpc.prior.mixture_distribution.probs))
        # stand-in plant: the model itself with nominal parameters
        state = model.step(state.view(1, -1), a_seq[0].view(1, -1)).view(-1).detach()

    for k, v in per.items():
        if v:
            g[k] = np.stack(v).reshape((n_ticks, -1) + v[0].shape) if k != "states_iter0" else np.stack(v)
    for k, v in tick.items():
        g["tick_" + k] = np.stack(v)
    g["kernel_kind"] = kernel_kind
    g["k2_bandwidth"] = float(k2_bandwidth)
    g["lik_kind"] = lik_kind
    g["roll_strategy"] = roll_strategy
    g["model_kind"] = model_kind
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **g)
    print("wrote", tag, {k: getattr(v, "shape", v) for k, v in g.items() if k in ("eps", "costs", "phi")})
    return g


# ---------------------------------------------------------------- MultiDISCO (MPPI mode: internal noise + step)
def run_disco(tag, seed=3):
    torch.manual_seed(seed)
    N, H, S = 4, 8, 16
    model = PendulumModel()
    ctrl = MultiDISCO(
        model.observation_space, model.action_space, H, N, S, temperature=0.7, ctrl_penalty=0.4,
        a_cov=1.5 ** 2 * torch.eye(1), inst_cost_fn=pend_inst_cost, term_cost_fn=pend_term_cost, params_sampling=None,
    )
    ctrl.a_mat = torch.randn(N, H, 1)
    g = dict(N=N, H=H, S=S, a_mat0=npf(ctrl.a_mat), sigma_a=1.5, temperature=0.7, a_reg=ctrl.a_reg, state=np.array([3.0, 0.0], np.float32))
    _REC.clear()
    costs, states, actions, omega, _ = ctrl.forward(torch.tensor([3.0, 0.0]), model)  # disco.py:348
    z = [r for r in _REC if tuple(r.shape) == (S, N, H, 1)]
    assert len(z) == 1
    g.update(z=npf(z[0]), costs=npf(costs), states=npf(states), actions=npf(actions), omega=npf(omega),
             a_mat1=npf(ctrl.a_mat), a_mix=npf(ctrl.a_mix))
    import copy

    for strat in ("argmax", "average"):
        c = copy.deepcopy(ctrl)
        out = c.step(strategy=strat, steps=2)  # disco.py:396
        g["step_%s_actions" % strat] = npf(out)
        g["step_%s_a_seq" % strat] = npf(c.a_seq)
        g["step_%s_a_mat" % strat] = npf(c.a_mat)
    c = copy.deepcopy(ctrl)
    ext = torch.randn(H, 1) * 3
    out = c.step(strategy="external", steps=1, ext_actions=ext)
    g.update(step_external_in=npf(ext), step_external_actions=npf(out), step_external_a_seq=npf(c.a_seq))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **g)
    print("wrote", tag)


# ---------------------------------------------------------------- MPF scenarios
def run_mpf(tag, model_kind, Mp, n_steps, log_space, bw, seed=5, optimizer="SGD", lr_override=None):
    torch.manual_seed(seed)
    if model_kind == "pendulum":
        model = PendulumModel(uncertain_params=("length", "mass"))
        x0 = dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])).sample([Mp])
        obs0 = torch.tensor([3.0, 0.0])
        action = torch.tensor(1.3)
        true_model = PendulumModel(length=0.9, mass=1.1)
        lr = 0.001  # pendulum_config.yaml:29
        obs1 = true_model.step(obs0.view(1, -1), action.view(1, 1)).view(-1)
    else:
        model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
        x0 = dist.Normal(2.0, 0.1).sample([Mp, 1]).clamp(min=1e-6)
        obs0 = torch.tensor([-9.0, -9.0, 0.5, -0.25])
        action = torch.tensor([4.0, -7.0])
        true_model = Particle(**PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(3.0))
        lr = 0.01  # particle_config.yaml:34
        obs1 = true_model.step(obs0, action)
    if log_space:
        x0 = x0.clamp(min=1e-6).log()
    if lr_override is not None:
        lr = lr_override
    lik = GaussianLikelihood(initial_obs=obs0, obs_std=0.1, model=model, log_space=log_space)
    if optimizer == "Adam":  # the class default of SVGD (svgd.py:115): state persists across optimize() calls (mpf.py:24)
        mpf = MPF(init_particles=x0.clone(), likelihood=lik, lr=lr, bw=bw, bw_scale=1.0)
        assert isinstance(mpf.optimizer, torch.optim.Adam)
    else:
        mpf = MPF(init_particles=x0.clone(), likelihood=lik, optimizer_class=torch.optim.SGD, lr=lr, bw=bw, bw_scale=1.0)
    g = dict(Mp=Mp, P=x0.shape[1], n_steps=n_steps, log_space=int(log_space), bw=bw, lr=lr, obs_std=0.1,
             x0=npf(x0), obs0=npf(obs0), obs1=npf(obs1), action=npf(action).reshape(-1), model_kind=model_kind, optimizer=optimizer)
    # one bare phi evaluation (mpf.py:40) on the conditioned likelihood
    lik.condition(action, obs1)
    g["phi0"] = npf(mpf.phi(bw))
    # undo the conditioning so optimize() conditions exactly once, as the sim loop does
    lik.loc = obs0
    lik.past_obs = None
    lik.past_action = None
    grads, bw_out = mpf.optimize(action, obs1, bw=bw, n_steps=n_steps)  # mpf.py:64
    g.update(x_final=npf(mpf.x), grad_norms=npf(grads), bw_out=float(bw_out),
             prior_means=npf(mpf.prior.component_distribution.base_dist.loc))
    # second tick from the updated filter
    action2 = action * 0.5
    if model_kind == "pendulum":
        obs2 = true_model.step(obs1.view(1, -1), action2.view(1, 1)).view(-1)
    else:
        obs2 = true_model.step(obs1, action2)
    grads2, _ = mpf.optimize(action2, obs2, bw=bw, n_steps=n_steps)
    g.update(action2=npf(action2).reshape(-1), obs2=npf(obs2), x_final2=npf(mpf.x), grad_norms2=npf(grads2))
    # prior density of the resulting GMM at a few probe points (the controller reads it through .log_prob / .sample)
    probe = torch.linspace(-1, 3, 7).view(-1, 1).expand(-1, x0.shape[1]).contiguous()
    g.update(probe=npf(probe), probe_log_prob=npf(mpf.prior.log_prob(probe)))
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **g)
    print("wrote", tag)


# ---------------------------------------------------------------- obstacle maps + collision lookups
def run_maps(tag):
    g = {}
    for preset, w in (("grid_4x4", 2.1), ("grid_3x3", 2.0), ("staggered_3-2-3", 2.0), ("staggered_4-3-4-3-4", 1.5),
                      ("grid_6x6", 1.2), ("single_centred", 3.0)):
        m = generate_obstacle_map([22, 22], get_obst_preset(preset, w), 0.1, map_type="direct")
        bm = m.map.astype(np.uint8)
        assert set(np.unique(bm)) <= {0, 1}
        g["map_%s_w%s" % (preset, str(w).replace(".", "p"))] = np.packbits(bm, axis=None)
        g["shape_%s" % preset] = np.array(bm.shape)
    m = generate_obstacle_map([22, 22], get_obst_preset("grid_4x4", 2.1), 0.1, map_type="direct")
    torch.manual_seed(11)
    pts = torch.cat([
        (torch.rand(4000, 2) - 0.5) * 26.0,  # incl. out-of-bounds points
        torch.tensor([[-11.0, -11.0], [11.0, 11.0], [10.99999, -10.99999], [0.0, 0.0], [-0.05, 0.05], [1e9, -1e9],
                      [4.95, 4.95], [4.949999, 7.05], [7.0500001, 7.05], [-7.05, -4.95]]),
    ])
    g["points"] = npf(pts)
    g["collisions"] = npf(m.get_collisions(pts))  # obstacle_map.py:64
    g["n_occupied_grid_4x4"] = int(m.map.sum())
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **g)
    print("wrote", tag, "occupied", g["n_occupied_grid_4x4"])


if __name__ == "__main__":
    # Pendulum, nominal dynamics (cfg1/cfg2 shape, scaled down)
    run_svmpc("pend_k1", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K1", n_iters=3, n_ticks=3, seed=0)
    run_svmpc("pend_k1_close", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K1", n_iters=2, n_ticks=1, seed=1, theta_shrink=0.05)
    run_svmpc("pend_k2", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K2", n_iters=3, n_ticks=2, seed=2)
    run_svmpc("pend_k1_params", "pendulum", N=12, H=9, S=8, M=4, kernel_kind="K1", n_iters=2, n_ticks=2, seed=3, params_kind="uniform2")
    run_svmpc("pend_k1_expcost", "pendulum", N=8, H=6, S=16, M=1, kernel_kind="K1", lik_kind="ExpectedCost", weighted_prior=True,
              n_iters=2, n_ticks=2, seed=4, alpha=0.05)
    run_svmpc("pend_k1_ctrlpen", "pendulum", N=8, H=6, S=16, M=1, kernel_kind="K1", n_iters=2, n_ticks=1, seed=6, ctrl_penalty=0.5)
    run_svmpc("pend_k1_mean", "pendulum", N=8, H=6, S=8, M=1, kernel_kind="K1", roll_strategy="mean", n_iters=1, n_ticks=2, seed=7)
    run_svmpc("pend_cfg1", "pendulum", N=32, H=15, S=128, M=1, kernel_kind="K1", n_iters=1, n_ticks=2, seed=8)
    # Particle (2-D point mass, obstacle grid, crash), log-mass from an MPF-style GMM
    run_svmpc("part_k1_gmm", "particle", N=8, H=12, S=8, M=4, kernel_kind="K1", weighted_prior=True, n_iters=2, n_ticks=2, seed=9,
              params_kind="logmass_gmm")
    run_svmpc("part_k2_gmm", "particle", N=8, H=12, S=8, M=4, kernel_kind="K2", weighted_prior=True, n_iters=2, n_ticks=1, seed=10,
              params_kind="logmass_gmm")
    run_svmpc("part_k2shared", "particle", N=8, H=12, S=8, M=4, kernel_kind="K2shared", n_iters=1, n_ticks=1, seed=12,
              params_kind="logmass_gmm")
    run_svmpc("part_k1_scalar", "particle", N=6, H=10, S=8, M=4, kernel_kind="K1", n_iters=1, n_ticks=1, seed=13,
              params_kind="scalar_normal")
    run_svmpc("part_k1_near_obst", "particle", N=8, H=40, S=16, M=2, kernel_kind="K1", n_iters=1, n_ticks=1, seed=14,
              params_kind="logmass_gmm", state0=[-5.2, -7.3, 4.0, 3.0])
    run_disco("disco_mppi")
    run_mpf("mpf_pend", "pendulum", Mp=10, n_steps=5, log_space=False, bw=0.08)
    run_mpf("mpf_part_log", "particle", Mp=12, n_steps=20, log_space=True, bw=0.5)
    run_maps("maps")
