"""Forward models with the reference's constructor signatures and attributes (dust/models/base.py, pendulum.py,
particle.py).  A model here is a DESCRIPTION: family id + parameters that `dust_amd.controllers.MultiDISCO` flattens into
`dust_config`; the batched rollouts run in the HIP kernels.  `step()` on a handful of states (the plant in a closed-loop
driver, not the hot path) is plain torch on the host."""
import math

import torch

from ..utils.obstacle_map import generate_obstacle_map, get_obst_preset
from ..utils.spaces import Box


class BaseModel:
    family = None

    def __init__(self, dt=0.05, params_dict=None, uncertain_params=None):
        assert dt > 0, "Delta t must be greater than zero."
        self._dt = dt
        self._params_dict = dict(params_dict or {})
        self._params_keys = uncertain_params

    @property
    def dt(self):
        return self._dt

    @property
    def params_dict(self):
        return self._params_dict

    @params_dict.setter
    def params_dict(self, d):
        self._params_dict = d

    @property
    def uncertain_params(self):
        return self._params_keys

    def params_to_dict(self, params):  # base.py:173-177
        return {key: params[:, idx].reshape(-1, 1) for (idx, key) in enumerate(self._params_keys)}

    def dict_to_params(self, params_dict):  # base.py:179-183
        return torch.cat([params_dict[key] for key in self._params_keys], dim=1)

    def _merged(self, params_dict):
        if params_dict is None:
            return self._params_dict
        merged = self._params_dict.copy()
        merged.update(params_dict)
        return merged


class PendulumModel(BaseModel):
    """dust/models/pendulum.py:9-108."""

    family = "pendulum"

    def __init__(self, g=9.8, mass=1.0, length=1.0, **kwargs):
        super().__init__(params_dict={"g": g, "mass": mass, "length": length}, **kwargs)
        self._max_speed, self._max_torque = 8.0, 2.0
        bounds = torch.tensor([float("inf"), self._max_speed])
        self._observation_space = Box(dim=2, low=-bounds, high=bounds, dtype=torch.float)
        self._action_space = Box(dim=1, low=-self._max_torque, high=self._max_torque, dtype=torch.float)

    observation_space = property(lambda self: self._observation_space)
    action_space = property(lambda self: self._action_space)

    def step(self, states, actions, params_dict=None):
        p = self._merged(params_dict)
        g, m, length = p["g"], p["mass"], p["length"]
        theta, theta_d = torch.as_tensor(states, dtype=torch.float).clone().chunk(2, dim=-1)
        acts = torch.as_tensor(actions, dtype=torch.float).clamp(min=-self._max_torque, max=self._max_torque)
        theta_d = theta_d + self.dt * (-3 * g / (2 * length) * (theta + math.pi).sin() + 3.0 / (m * length ** 2) * acts)
        theta_d = theta_d.clamp(-self._max_speed, self._max_speed)
        theta = theta + theta_d * self.dt
        return torch.cat((theta, theta_d), dim=-1)


class Particle(BaseModel):
    """dust/models/particle.py:11-334 (`render` is out of scope).  `deterministic=False` with a non-zero `noise_std` and
    `control_type="velocity"` run on the device through csrc/particle_general.hpp."""

    family = "particle"

    def __init__(self, mass=1.0, noise_std=torch.zeros(2), control_type="acceleration", cost_params=None, with_obstacle=False,
                 obst_preset=None, obst_width=None, obst_params=None, map_size=None, map_type=None, map_cell_size=None,
                 init_state=None, target_state=None, can_crash=False, max_speed=None, max_accel=None, verbose=False,
                 deterministic=False, euler_steps=1, **kwargs):
        super().__init__(params_dict={"mass": mass}, **kwargs)
        self._max_speed = float("inf") if max_speed is None else max_speed
        self._max_acc = float("inf") if max_accel is None else max_accel
        if control_type == "velocity":  # particle.py:41-48: a TWO-state model (x, y)
            bounds = torch.tensor([float("inf"), float("inf")])
            self._observation_space = Box(dim=2, low=-bounds, high=bounds, dtype=torch.float)
            self._action_space = Box(dim=2, low=-float(self._max_speed), high=float(self._max_speed), dtype=torch.float)
        elif control_type == "acceleration":
            bounds = torch.tensor([float("inf"), float("inf"), float(self._max_speed), float(self._max_speed)])
            self._observation_space = Box(dim=4, low=-bounds, high=bounds, dtype=torch.float)
            self._action_space = Box(dim=2, low=-float(self._max_acc), high=float(self._max_acc), dtype=torch.float)
        else:
            raise IOError('control_type "{}" not recognized'.format(control_type))  # particle.py:59-60
        self.target = torch.zeros(self._observation_space.dim) if target_state is None else torch.as_tensor(target_state, dtype=torch.float)
        self.dyn_std = noise_std
        self.init_state = None if init_state is None else torch.as_tensor(init_state)
        self.euler_steps = euler_steps
        self.control_type = control_type  # (set before init_cost_weights, which reads it - as particle.py:68-91 orders them)
        self.with_obstacle, self.can_crash = with_obstacle, can_crash
        self.map_cell_size, self.map_size = map_cell_size, map_size
        self.verbose, self.deterministic = verbose, deterministic
        self.init_cost_weights(cost_params)
        self.obst_map = None
        if self.with_obstacle:
            self.obst_params = get_obst_preset(obst_preset, obst_width)
            self.obst_map = generate_obstacle_map(map_size, self.obst_params, map_cell_size, map_type=map_type)

    observation_space = property(lambda self: self._observation_space)
    action_space = property(lambda self: self._action_space)

    def init_cost_weights(self, params):  # particle.py:292-326
        if params is None:
            params = dict.fromkeys(["w_qpos", "w_qvel", "w_qpos_T", "w_qvel_T", "w_ctrl", "w_obs"], 1.0)
        vel = self.control_type == "velocity"  # two-state model: position weights only (particle.py:307-322)
        self.w_state = torch.as_tensor([params["w_qpos"]] * 2 + ([] if vel else [params["w_qvel"]] * 2), dtype=torch.float)
        self.w_ctrl = torch.as_tensor([params["w_ctrl"]] * 2, dtype=torch.float)
        self.w_term = torch.as_tensor([params["w_qpos_T"]] * 2 + ([] if vel else [params["w_qvel_T"]] * 2), dtype=torch.float)
        self.w_obs = torch.as_tensor([params["w_obs"]], dtype=torch.float)

    def step(self, states, actions, params_dict=None):  # particle.py:117-166, plant-side
        (m,) = self._merged(params_dict).values()
        states = torch.as_tensor(states, dtype=torch.float)
        acts = torch.as_tensor(actions, dtype=torch.float).clone()
        if not self.deterministic:
            acts = acts + torch.as_tensor(self.dyn_std, dtype=torch.float) * torch.randn_like(acts)
        if self.control_type == "acceleration":
            acts = torch.clamp(acts / m, min=-self._max_acc, max=self._max_acc)
        else:  # particle.py:152-153
            acts = acts.clamp(min=-self._max_speed, max=self._max_speed)
        x_dot = torch.cat((states[..., 2:], acts), dim=-1)
        if self.can_crash and self.with_obstacle:
            mask = self.obst_map.get_collisions(states[..., 0:2]).unsqueeze(-1)
            nxt = states + x_dot * self.dt * (1 - mask)
        else:
            nxt = states + x_dot * self.dt
        nxt[..., -2:].clamp_(min=-self._max_speed, max=self._max_speed)
        return nxt

    # tagged cost callables: MultiDISCO recognises them and uses the fused HIP cost; calling them evaluates the same formula
    def default_inst_cost(self, states, actions=0, n_pol=0, debug=False):  # particle.py:170-198
        obst = self.w_obs * self.obst_map.get_collisions(states[..., 0:2]) if self.with_obstacle else 0.0
        d = states - self.target
        return (d * d * self.w_state).sum(-1) + (torch.mul(actions, actions) * self.w_ctrl).sum(-1) + obst

    def default_term_cost(self, states, n_pol=0, debug=False):  # particle.py:202-225
        obst = self.w_obs * self.obst_map.get_collisions(states[..., 0:2]) if self.with_obstacle else 0.0
        d = states - self.target
        return (d * d * self.w_term).sum(-1) + obst


class SkidSteerRobot(BaseModel):
    """dust/models/skid_steer_robot.py:9-122 (Kozlowski & Pazderski's simplified kinematic model).  State (x, y, theta, v, omega),
    action (right, left) wheel speeds [rot/s].  The reference ships no cost family and no demo for it; batched rollouts run on the
    device (csrc/skid.hpp) with `dust_amd.costs.QuadraticCost` as the cost family; `step` below is the plant-side host form."""

    family = "skid_steer"

    def __init__(self, delta_t, x_icr=0.2, wheel_radius=0.0625, axial_distance=0.475, min_wheel_speed=-0.5, max_wheel_speed=0.5, **kwargs):
        super().__init__(dt=delta_t, params_dict={"x_icr": x_icr, "wheel_radius": wheel_radius, "axial_distance": axial_distance}, **kwargs)
        self._observation_space = Box(dim=5, low=-float("inf"), high=float("inf"), dtype=torch.float)
        self._action_space = Box(dim=2, low=min_wheel_speed, high=max_wheel_speed, dtype=torch.float)

    observation_space = property(lambda self: self._observation_space)
    action_space = property(lambda self: self._action_space)

    def step(self, states, actions, params_dict=None):
        p = self._merged(params_dict)
        x_icr, radius, axial = p["x_icr"], p["wheel_radius"], p["axial_distance"]
        states = torch.as_tensor(states, dtype=torch.float)
        x, y, theta, _, _ = states.chunk(5, dim=1)
        right, left = torch.as_tensor(actions, dtype=torch.float).clone().chunk(2, dim=1)
        right = right.clamp(self.action_space.low[0], self.action_space.high[0])
        left = left.clamp(self.action_space.low[1], self.action_space.high[1])
        linear = (right + left) * math.pi * radius
        angular = (right - left) * 2 * math.pi * radius / axial
        forward = linear * self.dt
        lateral = -angular * x_icr * self.dt
        new_x = x + forward * torch.cos(theta) - lateral * torch.sin(theta)
        new_y = y + forward * torch.sin(theta) + lateral * torch.cos(theta)
        new_theta = theta + angular * self.dt
        return torch.cat([new_x, new_y, new_theta, linear.expand_as(x), angular.expand_as(x)], dim=1)
