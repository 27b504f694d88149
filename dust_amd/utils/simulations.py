"""Closed-loop drivers with the reference's entry points (dust/utils/simulations.py:13-260): `run_pendulum_simulation`
and `run_particle_episode`, same arguments, same per-tick call order, same result formats (the pendulum driver returns the
pandas frame demo/pendulum_example.py pickles to `data.pkl`; the particle driver returns the cumulated cost).

Everything on the hot path (svmpc.optimize / forward, controller.forward / step, mpf.optimize) runs on the MI355X through
the C ABI.  The PLANT is host code by nature (one state, one step per tick): the reference steps gym's `Pendulum-v0`,
which is not installable here, so the pendulum driver steps a `PendulumModel` carrying the episode's true (length, mass)
and gym's g = 10, max torque 2, max speed 8 (the same equations gym integrates); the particle driver steps a deep copy
of the controller's own `Particle`, as the reference does.  `render` is out of scope (plots)."""
from copy import deepcopy

import torch

from ..controllers.dual import DualSVMPC
from ..inference.likelihoods import ExponentiatedUtility
from ..inference.svmpc import SVMPC
from ..models import PendulumModel


def _nan(*shape):
    return torch.full(shape, float("nan"), dtype=torch.float)


def run_pendulum_simulation(init_state, init_policies, model_kwargs, dyn_dist, experiment_params, controller, use_exact_model=True,
                            use_svmpc=True, svmpc_kwargs=None, lik_kwargs=None, mpf=None, mpf_bw=None, mpf_steps=20, episodes=3,
                            steps=200, render=False, warm_up=1, verbose=False, steps_per_message=20):
    import pandas as pd

    if render:
        raise NotImplementedError("render: plotting is out of scope for the MI355X build")
    frames = []
    for ep in range(episodes):
        truth = experiment_params[ep]
        print("--- Starting iteration {} ---".format(ep + 1))
        print("parameters are: " + " and ".join("{} {:.2f}".format(k, float(v)) for k, v in truth.items()))
        if use_exact_model:
            model = PendulumModel(**truth, **model_kwargs)
        else:
            model = PendulumModel(length=dyn_dist.mean[0], mass=dyn_dist.mean[1], **model_kwargs)
        plant = PendulumModel(g=10.0, length=float(truth["length"]), mass=float(truth["mass"]))  # gym Pendulum-v0 stand-in
        state = torch.as_tensor(init_state, dtype=torch.float).reshape(1, -1)
        # the controller of this episode: same starting plan as SVMPC's particles, separate storage (simulations.py:60-63)
        sim_ctrl = deepcopy(controller)
        sim_ctrl.a_mat = init_policies.detach().clone()
        sim_ctrl.return_rollouts = False  # nothing below reads the sampled states / actions (no rendering)
        sim_svmpc = None
        if use_svmpc:
            if svmpc_kwargs is None or lik_kwargs is None:
                raise AssertionError("Need a Stein Optimizer and likelihood for dual svmpc simulation.")
            sim_svmpc = SVMPC(likelihood=ExponentiatedUtility(**lik_kwargs, controller=sim_ctrl, model=model), **svmpc_kwargs)
        sim_mpf, dyn_particles, dyn_bws = None, None, None
        ep_dyn_dist = dyn_dist
        if mpf is not None:
            sim_mpf = deepcopy(mpf)
            ep_dyn_dist = sim_mpf.prior
            dyn_particles = _nan(steps, *sim_mpf.x.size())
            dyn_bws = torch.zeros(steps)
        # the two inferences of the dual loop as one object (controllers/dual.py): forward() = optimize (+ forward once warmed up,
        # simulations.py:108-123), step() = the filter update (simulations.py:132-138)
        dual = DualSVMPC(sim_svmpc, sim_mpf, dyn_dist=ep_dyn_dist, mpf_bw=mpf_bw, mpf_steps=mpf_steps, warm_up=warm_up) if use_svmpc else None
        states, actions, costs = _nan(steps, sim_ctrl.dim_s), _nan(steps, sim_ctrl.dim_a), _nan(steps, 1)
        pol_particles = _nan(steps, sim_ctrl.n_pol, sim_ctrl.hz_len, sim_ctrl.dim_a)
        weights = _nan(steps, sim_ctrl.n_pol)
        action, cost = torch.zeros(sim_ctrl.dim_a), torch.zeros(1)
        for step in range(steps):
            if use_svmpc:
                a_seq, p_weights = dual.forward(state)
                action = a_seq[0]
                if p_weights is not None:
                    pol_particles[step] = dual.theta.detach().clone()
                    weights[step] = p_weights
            else:
                sim_ctrl.forward(state, model, ep_dyn_dist)
                action = sim_ctrl.step(strategy="average").flatten()
            actions[step] = action
            state = plant.step(state, torch.as_tensor(action, dtype=torch.float).clamp(-2.0, 2.0).reshape(1, -1)).reshape(1, -1)
            if sim_mpf is not None:
                if dual is not None:
                    _, bw = dual.step(action, state)
                else:
                    _, bw = sim_mpf.optimize(action.squeeze(), state, bw=mpf_bw, n_steps=mpf_steps)
                dyn_particles[step] = sim_mpf.x
                dyn_bws[step] = bw
            cost = sim_ctrl.inst_cost_fn(state.view(1, -1))
            if verbose and not step % steps_per_message:
                print("Step {0}: action taken {1:.2f}, cost {2:.2f}".format(step, float(action), float(cost)))
                print("Current state: theta={0[0]}, theta_dot={0[1]}".format(state.squeeze()))
            states[step] = state
            costs[step] = cost
        if verbose:
            print("Last step {0}: action taken {1:.2f}, cost {2:.2f}".format(steps - 1, float(action), float(cost)))
        # column set of the reference's frame (simulations.py:171-190); it hard-codes 200 rows, here: `steps`
        df = pd.DataFrame(index=list(range(steps)), data={
            "Cost": costs[:, 0].numpy(), "Position": states[:, 0].numpy(), "Speed": states[:, 1].numpy(), "Actions": actions[:, 0].numpy(),
            "Timestep": torch.arange(steps).numpy(), "Iteration": ep,
            "DynParticles": dyn_particles.tolist() if dyn_particles is not None else None,
            "DynBandwidths": None if dyn_bws is None else dyn_bws.numpy(),
            "PolParticles": pol_particles[..., 0, 0].tolist(), "Weights": weights.tolist(),
            "ExpParams": steps * [list(truth.values())],
        })
        df["AvgCumCost"] = (df["Cost"].cumsum(0) / (df["Timestep"] + 1)).round(2)
        frames.append(df)
    return pd.concat(frames, axis=0) if frames else pd.DataFrame()


def run_particle_episode(init_state, model, dyn_dist, controller, use_svmpc=True, warm_up=30, svmpc=None, load=0, steps=400,
                         render=False, save_path=None, mpf=None, mpf_bw=None, mpf_steps=20, verbose=False):
    """Point-mass navigation episode: the simulated system starts as a copy of `model`; after a quarter of the episode its
    mass grows by `load` (simulations.py:208-209).  Ends on a crash (cost inf), within 1.0 of the target, or after `steps`.
    `mpf` (optional, as particle_example.py:196-203 does inline): the dynamics filter updated after every plant step;
    pass `dyn_dist = mpf.prior` with it."""
    if render:
        raise NotImplementedError("render: plotting is out of scope for the MI355X build")
    system = deepcopy(model)
    controller.return_rollouts = False  # the sampled states only feed the (unsupported) renderer
    state = torch.as_tensor(init_state, dtype=torch.float).clone()
    cum_cost = 0
    for step in range(steps):
        if step == steps // 4:
            system.params_dict["mass"] = system.params_dict["mass"] + load
        if use_svmpc:
            svmpc.optimize(state, dyn_dist)
            if step < warm_up:
                action = torch.zeros(controller.dim_a)
            else:
                a_seq, _ = svmpc.forward(state, dyn_dist)
                action = a_seq[0]
        else:
            controller.forward(state, model, params_dist=dyn_dist)
            action = controller.step(strategy="argmax")
        new_state = system.step(state, action.squeeze())
        if mpf is not None:
            mpf.optimize(action.squeeze(), new_state, bw=mpf_bw, n_steps=mpf_steps)
        state = new_state
        cost = controller.inst_cost_fn(state.view(1, -1))
        cum_cost = cum_cost + cost
        if verbose and step % 10 == 0:
            print("step %3d  pos (%.2f, %.2f)  vel (%.2f, %.2f)  cost %.1f" % (step, *[float(v) for v in state], float(cost)))
        if system.with_obstacle and bool(system.obst_map.get_collisions(state[:2])):
            print("Crashed at step {}".format(step))
            return float("inf")
        if float((system.target - state).norm()) <= 1.0:
            break
    return cum_cost
