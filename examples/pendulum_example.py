#!/usr/bin/env python3
"""Closed-loop DuSt-MPC on the inverted pendulum with the MI355X backend.

Counterpart of the reference's demo/pendulum_example.py + dust/utils/simulations.py:run_pendulum_simulation: the same
objects are built with the same arguments and the same per-tick call order

    svmpc.optimize(state, dyn_dist) -> svmpc.forward(state, dyn_dist) -> plant step -> mpf.optimize(action, new_obs)

only the imports say `dust_amd` instead of `dust`, and the gym `Pendulum-v0` plant (not installable here) is replaced by
a PendulumModel with the episode's true (length, mass) and gym's g = 10.  `--config` accepts a yaml with the reference's
keys (demo/pendulum_config.yaml); the defaults below are that file's values.

    python examples/pendulum_example.py --steps 50 --case dual
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
from dust_amd.inference import MPF, SVMPC, ExponentiatedUtility, GaussianLikelihood, get_gmm  # noqa: E402
from dust_amd.kernels import RBF, RBFKernel, iid_mp  # noqa: E402
from dust_amd.models import PendulumModel  # noqa: E402

DEFAULTS = dict(
    sim_params=dict(episodes=1, steps=200, warm_up=0),
    exp_params=dict(init_state=[3.0, 0.0], horizon=30, n_particles=3, action_samples=128, params_samples=8, alpha=1, learning_rate=2.0,
                    bandwidth_scaling=1.0, ctrl_sigma=2, ctrl_dim=1, prior_sigma=2, weighted_prior=False, likelihood="ExponentiatedUtility",
                    kernel="rbf", mpf_n_particles=50, mpf_steps=20, mpf_log_space=False, mpf_learning_rate=0.001, mpf_bandwidth=None,
                    mpf_bandwidth_scaling=1.0, mpf_obs_std=0.1),
)


def inst_cost(states, controls=None, n_pol=1, debug=None):
    theta, theta_d = states.chunk(2, dim=1)
    return 50.0 * (theta.cos() - 1) ** 2 + 1.0 * theta_d ** 2


def term_cost(states, n_pol=1, debug=None):
    return inst_cost(states).squeeze()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--case", choices=["dual", "svmpc", "mppi", "disco"], default="dual")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    cfg = copy.deepcopy(DEFAULTS)
    if args.config:
        import yaml

        with open(args.config) as f:
            loaded = yaml.load(f, yaml.FullLoader)
        for k in ("sim_params", "exp_params"):
            cfg[k].update(loaded.get(k, {}))
    e, sim = cfg["exp_params"], cfg["sim_params"]
    steps = args.steps or sim["steps"]
    N = args.particles or e["n_particles"]
    torch.manual_seed(args.seed)
    H, S, M, alpha = e["horizon"], e["action_samples"], e["params_samples"], e["alpha"]

    env_model = PendulumModel()
    init_state = torch.as_tensor(e["init_state"]).clone()
    policies_prior = get_gmm(torch.randn(N, H, env_model.action_space.dim), torch.ones(N), e["prior_sigma"] ** 2 * torch.eye(e["ctrl_dim"]))
    init_policies = policies_prior.sample([N])
    dynamics_prior = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])), 1)
    true_params = dynamics_prior.sample()

    if args.case in ("mppi", "disco"):
        # the reference's "MPPI Baseline" / "DISCO" cases (pendulum_example.py:217-261): one policy, no SVGD; DISCO rolls the
        # sigma points of the dynamics prior out (unscented transform, yaml utf block: n = 2, alpha = 0.5)
        from dust_amd.utils.simulations import run_pendulum_simulation
        from dust_amd.utils.utf import MerweScaledUTF

        disco = args.case == "disco"
        controller = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H,
                                n_policies=1, action_samples=S, temperature=1 / alpha, a_cov=e["ctrl_sigma"] ** 2 * torch.eye(e["ctrl_dim"]),
                                inst_cost_fn=inst_cost, term_cost_fn=term_cost,
                                params_sampling=MerweScaledUTF(n=2, alpha=0.5) if disco else None, params_log_space=False)
        t0 = time.perf_counter()
        df = run_pendulum_simulation(init_state, init_policies[0].unsqueeze(0), dict(uncertain_params=("length", "mass")) if disco else {},
                                     dynamics_prior, [dict(length=float(true_params[0]), mass=float(true_params[1]))], controller,
                                     use_exact_model=False, use_svmpc=False, episodes=1, steps=steps, warm_up=sim["warm_up"])
        el = time.perf_counter() - t0
        print("%s: %d ticks, avg cost %.2f, %.1f ticks/s incl. host plumbing" % (args.case, steps, float(df["Cost"].mean()), steps / el))
        return
    kernel = RBFKernel() if e["kernel"] == "rbf" else iid_mp(base_kernel=RBF(bandwidth=-1), ctrl_dim=1, indep_controls=True)
    dual = args.case == "dual"
    controller = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H, n_policies=N,
                            action_samples=S, params_samples=M, temperature=1 / alpha, a_cov=e["ctrl_sigma"] ** 2 * torch.eye(e["ctrl_dim"]),
                            inst_cost_fn=inst_cost, term_cost_fn=term_cost, params_sampling=True if dual else None,
                            params_log_space=e["mpf_log_space"])
    model = PendulumModel(length=dynamics_prior.mean[0], mass=dynamics_prior.mean[1], uncertain_params=("length", "mass") if dual else None)
    sim_ctrl = copy.deepcopy(controller)
    sim_ctrl.a_mat = init_policies.detach().clone()
    likelihood = ExponentiatedUtility(alpha=alpha, n_samples=S, controller=sim_ctrl, model=model)
    svmpc = SVMPC(init_particles=init_policies, prior=policies_prior, likelihood=likelihood, kernel=kernel, n_particles=N,
                  bw_scale=e["bandwidth_scaling"], n_steps=1, optimizer_class=torch.optim.SGD, lr=e["learning_rate"],
                  weighted_prior=e["weighted_prior"])
    mpf, dyn_dist = None, None
    if dual:
        mpf_init = dynamics_prior.sample([e["mpf_n_particles"]])
        lik = GaussianLikelihood(initial_obs=init_state, obs_std=e["mpf_obs_std"], model=PendulumModel(uncertain_params=("length", "mass")),
                                 log_space=e["mpf_log_space"])
        mpf = MPF(init_particles=mpf_init, likelihood=lik, optimizer_class=torch.optim.SGD, lr=e["mpf_learning_rate"], bw=0.1,
                  bw_scale=e["mpf_bandwidth_scaling"])
        dyn_dist = mpf.prior

    plant = PendulumModel(g=10.0, length=float(true_params[0]), mass=float(true_params[1]))  # gym Pendulum-v0 stand-in
    state = init_state.unsqueeze(0)
    total_cost, t0 = 0.0, time.perf_counter()
    for step in range(steps):
        svmpc.optimize(state, dyn_dist)
        if step < sim["warm_up"]:
            action = torch.zeros(1)
        else:
            a_seq, p_weights = svmpc.forward(state, dyn_dist)
            action = a_seq[0]
        state = plant.step(state, action.clamp(-2.0, 2.0).view(1, 1))
        if mpf is not None:
            mpf.optimize(action.squeeze(), state.view(-1), bw=e["mpf_bandwidth"], n_steps=e["mpf_steps"])
        total_cost += float(inst_cost(state.view(1, -1)))
        if step % 20 == 0:
            extra = "" if mpf is None else "  (length, mass) ~ %s" % [round(float(v), 3) for v in mpf.x.mean(0)]
            print("step %3d  theta %+.3f  theta_dot %+.3f  action %+.2f%s" % (step, float(state[0, 0]), float(state[0, 1]), float(action), extra))
    el = time.perf_counter() - t0
    print("%s: %d ticks, avg cost %.2f, %.1f ticks/s incl. host plumbing (true length %.2f mass %.2f)"
          % (args.case, steps, total_cost / steps, steps / el, float(true_params[0]), float(true_params[1])))


if __name__ == "__main__":
    main()
