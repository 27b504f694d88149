"""`MultiDISCO` with the reference's constructor, attributes and `forward` / `step` signatures
(dust/controllers/disco.py:8-417, dust/controllers/base.py:7-66), executing on the MI355X through the C ABI.

The controller OWNS the device context (one dust_ctx): rollouts, costs, weights and - when an SVMPC is attached - the Stein
particles and prior live there.  `copy.deepcopy(controller)` clones the context (dust_clone), as the simulation loops
require (simulations.py:62, particle_example.py:166-175)."""
import copy

import numpy as np
import torch

from .. import _lib as L
from ..backend import Context
from ..costs import recognise
from ..utils.utf import MerweScaledUTF

Empty = torch.Size([])


class MultiDISCO:
    def __init__(self, observation_space, action_space, hz_len, n_policies, action_samples, temperature=1.0, ctrl_penalty=1.0,
                 a_cov=None, inst_cost_fn=None, term_cost_fn=None, params_sampling=True, params_samples=4, params_log_space=False,
                 init_actions=None, **kwargs):
        self.hz_len = hz_len
        self.dim_s, self.dim_a = observation_space.dim, action_space.dim
        self.min_a, self.max_a = action_space.low, action_space.high
        if inst_cost_fn is None and term_cost_fn is None:
            raise ValueError("Specify at least one cost function")
        self.inst_cost_fn, self.term_cost_fn = inst_cost_fn, term_cost_fn
        self.n_pol, self.n_actions = n_policies, action_samples
        self.temp = temperature
        self.a_reg = temperature * (1 - ctrl_penalty)
        self._ctrl_penalty = ctrl_penalty
        if a_cov is None:
            a_cov = torch.eye(self.dim_a)
        a_cov = torch.as_tensor(a_cov, dtype=torch.float)
        if not torch.equal(a_cov, torch.diag(torch.diag(a_cov))) and self.dim_a != 2:
            raise NotImplementedError("a full a_cov has a HIP kernel for dim_a = 2 (no CPU fallback)")
        self.a_dist = torch.distributions.multivariate_normal.MultivariateNormal(torch.zeros(self.dim_a), a_cov)
        self.a_pre = torch.inverse(a_cov)
        self._a_seq = torch.zeros((hz_len, self.dim_a))
        if init_actions is None:
            self._a_mat = torch.zeros(n_policies, hz_len, self.dim_a)
        else:
            assert init_actions.shape == (n_policies, hz_len, self.dim_a), "Initial actions shape mismatch."
            self._a_mat = init_actions.clone()
        self._a_mix = torch.ones(n_policies)
        self._params_log_space = params_log_space
        self._tf = None
        # forward() returns (costs, states, actions, omega, params_log_p) like the reference; states [M,S,N,H+1,ds] and
        # actions are tens of MB per call at demo sizes.  Callers that ignore them (the closed-loop drivers) set this False:
        # the rollout kernel then neither stores nor copies them (states / actions come back as None).
        self.return_rollouts = True
        if params_sampling is False or params_sampling is None or params_sampling == "none":
            self.n_params, self._sampling = 1, False
        elif params_sampling is True:
            self.n_params, self._sampling = params_samples, True
        elif isinstance(params_sampling, MerweScaledUTF):  # disco.py:124-131: sigma-point rollouts ("DISCO" case)
            assert self._params_log_space is False, "Distribution must not be on log space if using UTF."
            self.n_params, self._sampling = 1, True
            self._tf = params_sampling
        else:
            raise ValueError("Invalid value for 'params_sampling': {}".format(params_sampling))
        self.n_rollouts = self.n_params * self.n_actions * self.n_pol
        self._ctx = None
        self._ctx_key = None
        self._svmpc_cfg = {}
        self._device = kwargs.get("device", 0)
        self._seed = kwargs.get("seed", 0)
        # Reproducible runs: an object whose next_eps() / next_params() / next_ctrl_noise() return the next recorded draw (or None:
        # draw as usual) - policy noise [S,N,H,da] per SVGD step, dynamics samples [M,P] per sampling call, control-channel noise
        # [H, M*S*N, da] per rollout launch (Particle(deterministic=False), particle.py:145-148).  None (default): every draw is
        # fresh - policy / control noise from the device Philox stream, dynamics samples from params_dist.  The parity tests replay
        # the reference's own recorded draws through it (tests/helpers.py RecordedDraws).
        self.draw_source = None

    # ------------------------------------------------------------------ context management
    def _config(self, model, params_dist):
        if model.family not in ("pendulum", "particle", "skid_steer"):
            raise NotImplementedError("no rollout kernel family for %s" % type(model).__name__)
        chol = torch.linalg.cholesky(self.a_dist.covariance_matrix).diag()
        sigma = self.a_dist.covariance_matrix.diag().sqrt()  # svmpc.py:107-111
        cfg = dict(model=model.family, N=self.n_pol, S=self.n_actions, M=self._tf.pts if self._tf is not None else self.n_params, H=self.hz_len,
                   temperature=float(self.temp), ctrl_penalty=float(self._ctrl_penalty), alpha=1.0 / float(self.temp),
                   chol_a=chol.numpy(), sigma_a=sigma.numpy(), a_pre=self.a_pre.diag().numpy(),
                   a_cov=self.a_dist.covariance_matrix.numpy(),  # (a full matrix takes the off-diagonal path: disco.py:91-98)
                   min_a=self.min_a.numpy(), max_a=self.max_a.numpy(), device=self._device, seed=self._seed, dt=model.dt,
                   params_log_space=bool(self._params_log_space), sampling=self._sampling)
        if self._sampling:
            if self._tf is not None:
                self._scalar_event = False
            elif params_dist is not None:
                self._scalar_event = params_dist.event_shape == Empty
            elif getattr(self, "_scalar_event", None) is None:
                raise ValueError("params_sampling is on but no params_dist was given")
            cfg["uncertain_params"] = tuple(model.uncertain_params)
            cfg["params_scalar_event"] = self._scalar_event
        pd = model.params_dict
        for k in ("x_icr", "wheel_radius", "axial_distance"):  # SkidSteerRobot (skid_steer_robot.py:40-44)
            if k in pd:
                cfg[k] = float(pd[k])
        for k in ("g", "mass", "length"):
            if k in pd:
                v = pd[k]
                cfg[k] = float(v)
                if k == "mass" and isinstance(v, torch.Tensor):
                    cfg["mass_0dim"] = True
        if model.family == "particle":
            cfg.update(max_speed=float(model._max_speed), max_accel=float(model._max_acc), can_crash=bool(model.can_crash),
                       with_obstacle=bool(model.with_obstacle), cell_size=float(model.map_cell_size or 0.1),
                       # particle.py:13-31, 145-153: control-channel noise and the control type reach the device rollouts
                       control_type=str(model.control_type), deterministic=bool(model.deterministic),
                       noise_std=tuple(float(v) for v in torch.as_tensor(model.dyn_std, dtype=torch.float).reshape(-1).expand(2)))
        cfg.update(recognise(model, self.inst_cost_fn, self.term_cost_fn))
        cfg.update(self._svmpc_cfg)
        return cfg

    def _ensure_ctx(self, model, params_dist=None):
        cfg = self._config(model, params_dist)
        key = repr(sorted((k, np.asarray(v).tolist() if not isinstance(v, (str, bool, int, float, tuple)) else v) for k, v in cfg.items()))
        if self._ctx is not None and key == self._ctx_key:
            return self._ctx
        old = self._ctx
        state = None
        if old is not None:  # configuration changed: carry the state over
            state = dict(theta=old.get_theta(), prior=old.get_prior(), a_mat=old.get_a_mat(), a_seq=old.get_a_seq())
            old.close()
        grid = model.obst_map.map.astype(np.float32) if getattr(model, "obst_map", None) is not None else None
        self._ctx = Context(grid=grid, **cfg)
        self._ctx_key = key
        if self._tf is not None:
            self._ctx.set_param_weights(self._tf.loc_weights.numpy())
        if state is None:
            self._ctx.set_a_mat(self._a_mat.numpy())
            self._ctx.set_a_seq(self._a_seq.numpy())
        else:
            self._ctx.set_theta(state["theta"])
            self._ctx.set_prior(*state["prior"])
            self._ctx.set_a_mat(state["a_mat"])
            self._ctx.set_a_seq(state["a_seq"])
        return self._ctx

    def __deepcopy__(self, memo):
        new = copy.copy(self)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_ctx":
                new._ctx = None if v is None else v.clone()
            elif k in ("inst_cost_fn", "term_cost_fn"):
                setattr(new, k, v)
            else:
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    # ------------------------------------------------------------------ reference attributes backed by the device
    @property
    def a_mat(self):
        return torch.from_numpy(self._ctx.get_a_mat()) if self._ctx is not None else self._a_mat

    @a_mat.setter
    def a_mat(self, v):
        self._a_mat = torch.as_tensor(v, dtype=torch.float).detach().clone()
        if self._ctx is not None:
            self._ctx.set_a_mat(self._a_mat.numpy())

    @property
    def a_mix(self):
        return torch.from_numpy(self._ctx.get_a_mix()) if self._ctx is not None else self._a_mix

    @property
    def a_seq(self):
        return torch.from_numpy(self._ctx.get_a_seq()) if self._ctx is not None else self._a_seq

    @a_seq.setter
    def a_seq(self, v):
        self._a_seq = torch.as_tensor(v, dtype=torch.float).detach().clone()
        if self._ctx is not None:
            self._ctx.set_a_seq(self._a_seq.numpy())

    # ------------------------------------------------------------------ disco.py:348-394
    def _sigma_params(self, params_dist):
        """Sigma points of the parameter distribution (disco.py:238-251) and their weighted log-probability (288-291)."""
        try:
            cov, mean = params_dist.covariance_matrix, params_dist.mean
        except AttributeError:
            cov, mean = params_dist.variance.diag(), params_dist.mean
        sp = self._tf.compute_sigma_points(mean, cov).T.contiguous()  # [pts][P]
        lp = params_dist.log_prob(sp) @ self._tf.loc_weights
        return sp.reshape(1, self._tf.pts, -1).numpy(), lp.expand(self.n_actions, self.n_pol)

    def _sample_params(self, params_dist, n_sets=1):
        if not self._sampling:
            return None, None
        if self._tf is not None:
            sp, lp = self._sigma_params(params_dist)
            return np.repeat(sp, n_sets, axis=0), lp
        ps, lps = [], []
        for _ in range(n_sets):
            rec = self._recorded("params")  # recorded dynamics samples (draw_source), else a fresh draw
            p = params_dist.sample([self.n_params]) if rec is None else torch.as_tensor(rec, dtype=torch.float)
            lps.append(params_dist.log_prob(p))
            ps.append(p.reshape(self.n_params, -1))
        return torch.stack(ps).numpy(), lps[-1]

    def _recorded(self, kind):
        src = self.draw_source
        return None if src is None else getattr(src, "next_" + kind)()

    def _feed_ctrl_noise(self, ctx, n_sets=1):
        """Recorded control-channel noise for the next n_sets rollout launches, when the draw source has it."""
        if self.draw_source is None or not ctx.cfg.ctrl_noise:
            return
        rec = [self._recorded("ctrl_noise") for _ in range(n_sets)]
        if rec and rec[0] is not None:
            ctx.set_ctrl_noise(np.stack([np.asarray(r, np.float32) for r in rec]))

    def forward(self, state, model, params_dist=None, ext_actions=None, debug=False):
        ctx = self._ensure_ctx(model, params_dist)
        state = torch.as_tensor(state, dtype=torch.float).reshape(-1)
        params, params_log_p = self._sample_params(params_dist)
        acts = None if ext_actions is None else torch.as_tensor(ext_actions, dtype=torch.float).numpy()
        want = bool(self.return_rollouts)
        self._feed_ctrl_noise(ctx)
        costs, states, actions, omega = ctx.disco_forward(state.numpy(), acts, None if params is None else params[0],
                                                          want_states=want, want_actions=want)
        if not want:
            return torch.from_numpy(costs), None, None, torch.from_numpy(omega), params_log_p
        actions = torch.from_numpy(actions).unsqueeze(0).expand(self.n_params, -1, -1, -1, -1)
        return torch.from_numpy(costs), torch.from_numpy(states), actions, torch.from_numpy(omega), params_log_p

    def step(self, strategy="argmax", steps=1, ext_actions=None):  # disco.py:396-417
        if strategy not in ("argmax", "average", "external") or (strategy == "external" and ext_actions is None):
            raise ValueError("Invalid value for strategy.")
        if self._ctx is None:
            raise RuntimeError("step() before the first forward(): no rollouts have been evaluated yet")
        ext = None if ext_actions is None else torch.as_tensor(ext_actions, dtype=torch.float).numpy()
        try:
            return torch.from_numpy(self._ctx.disco_step(strategy, steps, ext))
        except L.DustError as e:
            raise ValueError(str(e))
