#!/usr/bin/env python3
"""Point-mass navigation through the 4x4 obstacle grid with DuSt-MPC on the MI355X backend.

Counterpart of the reference's demo/particle_example.py: the same objects with the same arguments (only the imports say
`dust_amd`), one episode per `episodes`, the mass of the simulated system grows by `extra_load` after a quarter of the
episode, the dynamics filter (MPF over the unknown mass, log space) is updated after every plant step, and the episode
ends on a crash, at the target or after `steps`.  `--config` takes a yaml with the reference's keys
(demo/particle_config.yaml); the defaults below are that file's values.  Plot / gif output is out of scope.

    python examples/particle_example.py --steps 120
"""
import argparse
import copy
import os
import sys
import time

import torch
import torch.distributions as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dust_amd.controllers import MultiDISCO  # noqa: E402
from dust_amd.inference import MPF, SVMPC, GaussianLikelihood, get_gmm  # noqa: E402
from dust_amd.inference import likelihoods  # noqa: E402
from dust_amd.kernels import RBF, RBFKernel, iid_mp  # noqa: E402
from dust_amd.models import Particle  # noqa: E402
from dust_amd.utils.simulations import run_particle_episode  # noqa: E402

DEFAULTS = dict(
    sim_params=dict(warm_up=5, steps=10, episodes=1),
    exp_params=dict(horizon=40, n_particles=6, action_samples=64, params_samples=4, alpha=1, learning_rate=100, bandwidth_scaling=1.0,
                    ctrl_sigma=5, ctrl_dim=2, likelihood="ExponentiatedUtility", sampling=True, kernel="rbf", use_svmpc=True, use_mpf=True,
                    prior_sigma=5, weighted_prior=True, dyn_prior="Normal", dyn_prior_arg1=2, dyn_prior_arg2=0.1, extra_load=1.0,
                    mpf_n_particles=50, mpf_steps=20, mpf_log_space=True, mpf_learning_rate=0.01, mpf_bandwidth=0.5,
                    mpf_bandwidth_scaling=1.0, mpf_obs_std=0.1),
    env_params=dict(dt=0.015, control_type="acceleration", noise_std=[0.1, 0.1], init_state=[-9.0, -9.0, 0, 0], target_state=[9.0, 9.0, 0, 0],
                    can_crash=True, with_obstacle=True, deterministic=True,
                    cost_params=dict(w_qpos=0.5, w_qvel=0.25, w_ctrl=0.2, w_obs=1.0e6, w_qpos_T=1.0e3, w_qvel_T=0.1),
                    obst_preset="grid_4x4", obst_width=2.1, max_speed=5, max_accel=10, map_cell_size=0.1, map_size=[22, 22], map_type="direct"),
)


def build(cfg):
    e, env = cfg["exp_params"], cfg["env_params"]
    N, H, da = e["n_particles"], e["horizon"], e["ctrl_dim"]
    state = torch.as_tensor(env["init_state"], dtype=torch.float).clone()
    policies_prior = get_gmm(torch.randn(N, H, da), torch.ones(N), e["prior_sigma"] ** 2 * torch.eye(da))
    init_policies = policies_prior.sample([N])
    dynamics_prior = getattr(dist, e["dyn_prior"])(e["dyn_prior_arg1"], e["dyn_prior_arg2"])
    system_kwargs = dict(uncertain_params=["mass"], mass=dynamics_prior.mean)
    model = Particle(**env, **system_kwargs)
    controller = MultiDISCO(model.observation_space, model.action_space, H, N, e["action_samples"], temperature=1 / e["alpha"],
                            a_cov=e["ctrl_sigma"] ** 2 * torch.eye(da), params_sampling=e["sampling"], params_samples=e["params_samples"],
                            params_log_space=e["mpf_log_space"], inst_cost_fn=model.default_inst_cost, term_cost_fn=model.default_term_cost)
    if e["kernel"] == "message_passing":
        kernel = iid_mp(base_kernel=RBF(bandwidth=-1), ctrl_dim=2, indep_controls=True)
    elif e["kernel"] == "rbf":
        kernel = RBFKernel()
    else:
        raise ValueError("Kernel type '{}' is not valid.".format(e["kernel"]))
    lik = getattr(likelihoods, e["likelihood"])(e["alpha"], controller=controller, model=model, n_samples=e["action_samples"])
    svmpc = SVMPC(init_particles=init_policies.detach().clone(), prior=policies_prior, likelihood=lik, kernel=kernel, n_particles=N,
                  bw_scale=e["bandwidth_scaling"], n_steps=1, optimizer_class=torch.optim.SGD, lr=e["learning_rate"],
                  weighted_prior=e["weighted_prior"])
    mpf_init = dynamics_prior.sample([e["mpf_n_particles"], 1]).clamp(min=1e-6)
    mpf_init = mpf_init.log() if e["mpf_log_space"] else mpf_init
    dyn_lik = GaussianLikelihood(initial_obs=state, obs_std=e["mpf_obs_std"], model=model, log_space=e["mpf_log_space"])
    mpf = MPF(init_particles=mpf_init, likelihood=dyn_lik, optimizer_class=torch.optim.SGD, lr=e["mpf_learning_rate"],
              bw=(2 * e["dyn_prior_arg2"]) ** 1 / 2, bw_scale=e["mpf_bandwidth_scaling"])  # operator precedence as in the reference
    return state, model, controller, svmpc, mpf, dynamics_prior


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--episodes", type=int, default=None)
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args(argv)
    cfg = copy.deepcopy(DEFAULTS)
    if args.config:
        import yaml

        with open(args.config) as f:
            loaded = yaml.load(f, yaml.FullLoader)
        for k in cfg:
            cfg[k].update(loaded.get(k, {}))
    if args.particles:
        cfg["exp_params"]["n_particles"] = args.particles
    sim, e = cfg["sim_params"], cfg["exp_params"]
    steps = args.steps or sim["steps"]
    torch.manual_seed(args.seed)
    state, base_model, base_controller, base_svmpc, base_mpf, dynamics_prior = build(cfg)
    results = []
    for ep in range(args.episodes or sim["episodes"]):
        model, svmpc = copy.deepcopy(base_model), copy.deepcopy(base_svmpc)  # handles survive deepcopy (particle_example.py:166-175)
        controller = svmpc.likelihood.controller
        mpf = copy.deepcopy(base_mpf) if e["use_mpf"] else None
        dyn_dist = mpf.prior if mpf is not None else dynamics_prior
        t0 = time.perf_counter()
        cost = run_particle_episode(state, model, dyn_dist, controller, use_svmpc=e["use_svmpc"], warm_up=sim["warm_up"], svmpc=svmpc,
                                    load=e["extra_load"], steps=steps, mpf=mpf, mpf_bw=e["mpf_bandwidth"], mpf_steps=e["mpf_steps"],
                                    verbose=args.verbose)
        el = time.perf_counter() - t0
        mass = None
        if mpf is not None:
            x = mpf.x
            mass = float((x.exp() if e["mpf_log_space"] else x).mean())
        results.append(float(cost))
        print("episode %d: cumulated cost %.4g in %.2f s%s" % (ep, float(cost), el, "" if mass is None else ", mass estimate %.3f" % mass))
    return results


if __name__ == "__main__":
    main()
