"""CPU ORACLE for the SVGD-MPC hot path (TEST INFRASTRUCTURE - not product code).

`oracle/dust_oracle.c` is a plain-C restatement of the reference's per-tick algorithm (every function cites the
reference file:line it follows); this module builds it with gcc and exposes it through ctypes + numpy.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package.  The
product (`dust_amd`, libdust_amd.so) never does - it fails loudly when its HIP library is missing.

Pinned against tests/golden/*.npz (vectors produced by the reference itself, tests/golden/make_golden.py) in
tests/test_oracle_golden.py.  Unpinned third-party boundaries: gpytorch RBFKernel (K1) and KDEpy silvermans_rule.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdust_oracle.so")

MODEL_PENDULUM, MODEL_PARTICLE = 0, 1
LIK_EXP_UTILITY, LIK_EXPECTED_COST = 0, 1
ROLL_REPEAT, ROLL_MEAN = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "dust_oracle.c")
    hdr = os.path.join(_HERE, "dust_oracle.h")
    if (not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=c11", "-D_GNU_SOURCE", "-o", _SO, src, "-lm"]
    subprocess.run(cmd, check=True)
    return _SO


class _Param(C.Structure):
    _fields_ = [("is_tensor", C.c_int), ("col", C.c_int), ("value", C.c_double)]


class _Cfg(C.Structure):
    _fields_ = [
        ("model", C.c_int), ("N", C.c_int), ("S", C.c_int), ("M", C.c_int), ("H", C.c_int), ("da", C.c_int),
        ("ds", C.c_int), ("P", C.c_int), ("params_interleave", C.c_int), ("params_log_space", C.c_int),
        ("dt", C.c_double),
        ("g", _Param), ("mass", _Param), ("length", _Param),
        ("max_torque", C.c_double), ("max_speed_pend", C.c_double), ("w_cos", C.c_double), ("w_vel", C.c_double),
        ("pmass", _Param), ("mass_is_0dim_tensor", C.c_int),
        ("max_speed", C.c_float), ("max_acc", C.c_float), ("can_crash", C.c_int), ("with_obstacle", C.c_int),
        ("cell_size", C.c_double), ("nx", C.c_int), ("ny", C.c_int), ("off_x", C.c_float), ("off_y", C.c_float),
        ("grid", C.POINTER(C.c_float)),
        ("target", C.c_float * 4), ("w_state", C.c_float * 4), ("w_term", C.c_float * 4), ("w_ctrl", C.c_float * 2),
        ("w_obs", C.c_float),
        ("velocity_ctrl", C.c_int), ("dyn_std", C.c_float * 2), ("ctrl_noise", C.POINTER(C.c_float)),
        ("full_cov", C.c_int), ("chol_a_full", C.c_float * 3), ("a_pre_full", C.c_float * 3), ("chol_p_full", C.c_float * 3),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def grid_4x4_map(width=2.1, cell=0.1, size=(22, 22)):
    """Occupancy grid of the `grid_4x4` preset + border walls (obstacle_map.py:155-175, 249-361; obstacle.py:57-69)."""
    from math import ceil

    nx, ny = ceil(size[0] / cell), ceil(size[1] / cell)
    m = np.zeros((nx, ny), np.float32)
    ox, oy = int(nx / 2), int(ny / 2)

    def add(cx, cy, w, h):
        cx, cy = int(cx), int(cy)  # Obstacle.__init__ truncates centres to int (obstacle.py:14-15)
        wc, hc = ceil(w / cell), ceil(h / cell)
        c_x, c_y = ceil(cx / cell), ceil(cy / cell)
        xs, xe = c_x - ceil(wc / 2.0) + ox, c_x + ceil(wc / 2.0) + ox
        ys, ye = c_y - ceil(hc / 2.0) + oy, c_y + ceil(hc / 2.0) + oy
        m[xs:xe, ys:ye] = 1  # raw (possibly negative) indices: numpy slice semantics are part of the behaviour

    s = 4
    for cy in (s * 3 / 2, s / 2, -s / 2, -s * 3 / 2):
        for cx in (-s * 3 / 2, -s / 2, s / 2, s * 3 / 2):
            add(cx, cy, width, width)
    xlim = (-cell * nx / 2, cell * nx / 2)
    ylim = (-cell * ny / 2, cell * ny / 2)
    for lim in xlim:
        add(lim, 0, 4 * cell, ylim[1] - ylim[0])
    for lim in ylim:
        add(0, lim, xlim[1] - xlim[0], 4 * cell)
    return m


class Oracle:
    """Holds an orc_cfg and forwards to the C functions with numpy arrays."""

    def __init__(self, model="pendulum", N=1, S=1, M=1, H=1, uncertain_params=None, params_scalar_event=False,
                 params_log_space=False, dt=None, g=9.8, mass=1.0, length=1.0, w_cos=50.0, w_vel=1.0,
                 grid=None, cell_size=0.1, max_speed=5.0, max_accel=10.0, can_crash=True, with_obstacle=True,
                 target=(9.0, 9.0, 0.0, 0.0), w_state=(0.5, 0.5, 0.25, 0.25), w_term=(1e3, 1e3, 0.1, 0.1),
                 w_ctrl=(0.2, 0.2), w_obs=1e6, mass_0dim=False, control_type="acceleration", noise_std=(0.0, 0.0), a_cov=None, p_cov=None):
        c = _Cfg()
        self.model = model
        c.model = MODEL_PENDULUM if model == "pendulum" else MODEL_PARTICLE
        c.N, c.S, c.M, c.H = N, S, M, H
        vel = model != "pendulum" and control_type == "velocity"  # particle.py:41-48: a two-state model
        c.da, c.ds = (1, 2) if model == "pendulum" else ((2, 2) if vel else (2, 4))
        c.velocity_ctrl = int(vel)
        c.dyn_std[:] = [float(v) for v in np.broadcast_to(np.asarray(noise_std, np.float32).reshape(-1), (2,))]
        up = list(uncertain_params) if uncertain_params else []
        c.P = max(len(up), 1)
        c.params_interleave = int(params_scalar_event)
        c.params_log_space = int(params_log_space)
        c.dt = (0.05 if model == "pendulum" else 0.015) if dt is None else dt

        def par(name, value):
            if name in up:
                return _Param(1, up.index(name), float(value))
            return _Param(2 if mass_0dim and name == "mass" else 0, 0, float(value))

        c.g, c.mass, c.length = par("g", g), par("mass", mass), par("length", length)
        c.max_torque, c.max_speed_pend, c.w_cos, c.w_vel = 2.0, 8.0, w_cos, w_vel
        c.pmass = par("mass", mass)
        c.max_speed, c.max_acc = max_speed, max_accel
        c.can_crash, c.with_obstacle = int(can_crash), int(with_obstacle and model != "pendulum")
        c.cell_size = cell_size
        if model != "pendulum":
            self._grid = _f(grid_4x4_map() if grid is None else grid)
            c.nx, c.ny = self._grid.shape
            c.off_x, c.off_y = int(c.nx / 2), int(c.ny / 2)
            c.grid = _p(self._grid)
        pad4 = lambda v: (list(v) + [0.0] * 4)[:4]
        c.target[:] = pad4(target)
        c.w_state[:] = pad4(w_state)
        c.w_term[:] = pad4(w_term)
        c.w_ctrl[:] = w_ctrl
        c.w_obs = w_obs
        if a_cov is not None or p_cov is not None:  # full 2 x 2 covariances (disco.py:91-98, svgd.py:84-89), fp32 as torch factors them
            import torch

            ac = torch.eye(2) if a_cov is None else torch.as_tensor(np.asarray(a_cov, np.float32))
            pc = torch.eye(2) if p_cov is None else torch.as_tensor(np.asarray(p_cov, np.float32))
            la, lp, ap = torch.linalg.cholesky(ac), torch.linalg.cholesky(pc), torch.inverse(ac)
            c.full_cov = 1
            c.chol_a_full[:] = [float(la[0, 0]), float(la[1, 0]), float(la[1, 1])]
            c.chol_p_full[:] = [float(lp[0, 0]), float(lp[1, 0]), float(lp[1, 1])]
            c.a_pre_full[:] = [float(ap[0, 0]), float(ap[0, 1]), float(ap[1, 1])]
        self.c = c
        self.D = H * c.da

    # -- a1
    def sample_actions(self, theta, eps, chol_a):
        theta, eps, chol_a = _f(theta), _f(eps), _f(chol_a)
        out = np.empty_like(eps)
        lib().orc_sample_actions(C.byref(self.c), _p(theta), _p(eps), _p(chol_a), _p(out))
        return out

    # -- a2..a5
    def rollout_cost(self, state, actions, params=None, a_reg=0.0, a_mat=None, a_seq=None, a_pre_diag=None, want_states=False,
                     ctrl_noise=None):
        """ctrl_noise: recorded control-channel draws [H][M*S*N][da] of Particle(deterministic=False) (particle.py:145-148), or None."""
        c = self.c
        cz = None if ctrl_noise is None else _f(ctrl_noise).reshape(c.H, c.M * c.S * c.N, c.da)
        c.ctrl_noise = _p(cz)
        state, actions = _f(state).reshape(-1), _f(actions)
        params = None if params is None else _f(params).reshape(c.M, -1)
        costs = np.empty((c.S, c.N), np.float32)
        states = np.empty((c.M, c.S, c.N, c.H + 1, c.ds), np.float32) if want_states else None
        a_mat = None if a_mat is None else _f(a_mat)
        a_seq = _f(np.zeros(self.D) if a_seq is None else a_seq)
        a_pre = _f(np.ones(c.da) if a_pre_diag is None else a_pre_diag)
        lib().orc_rollout_cost(C.byref(c), _p(state), _p(actions), _p(params), C.c_float(a_reg), _p(a_mat), _p(a_seq),
                               _p(a_pre), _p(states), _p(costs))
        c.ctrl_noise = None
        return (costs, states) if want_states else costs

    def rollout_cost_ut(self, state, actions, sigma_points, loc_weights, a_reg=0.0, a_mat=None, a_seq=None, a_pre_diag=None):
        """Unscented-transform rollouts (disco.py:211-292, 312-323): sigma_points [M][P], loc_weights [M] (self.c.M = M)."""
        c = self.c
        st, act = _f(state), _f(actions)
        sp, w = _f(sigma_points), _f(loc_weights)
        assert sp.shape == (c.M, c.P) and w.shape == (c.M,)
        costs = np.empty((c.S, c.N), np.float32)
        lib().orc_rollout_cost_ut(C.byref(c), _p(st), _p(act), _p(sp), _p(w), C.c_float(a_reg),
                                  _p(None if a_mat is None else _f(a_mat)), _p(None if a_seq is None else _f(a_seq)),
                                  _p(None if a_pre_diag is None else _f(a_pre_diag)), _p(costs))
        return costs

    # -- a6
    def disco_weights(self, costs, actions, eps_base, temp, a_mat):
        c = self.c
        costs, actions, a_seq = _f(costs), _f(actions), _f(eps_base)
        per_policy = int(a_seq.size == c.N * self.D)
        a_mat = _f(a_mat).copy()
        omega = np.empty((c.S, c.N), np.float32)
        a_mix = np.empty(c.N, np.float32)
        lib().orc_disco_weights(C.byref(c), _p(costs), _p(actions), _p(a_seq), C.c_int(per_policy), C.c_float(temp), _p(omega), _p(a_mat), _p(a_mix))
        return omega, a_mat, a_mix

    @staticmethod
    def log_mix(weights):
        w = _f(weights)
        out = np.empty_like(w)
        lib().orc_log_mix(C.c_int(w.size), _p(w), _p(out))
        return out

    # -- a9
    def score(self, theta, mu, mix_weights, sigma_p, costs, actions, alpha, sigma_a):
        c = self.c
        theta, mu, costs, actions = _f(theta), _f(mu), _f(costs), _f(actions)
        logmix = self.log_mix(mix_weights)
        sp, sa = _f(np.broadcast_to(sigma_p, (c.da,))), _f(np.broadcast_to(sigma_a, (c.da,)))
        gl, gp, sc = (np.empty((c.N, c.H, c.da), np.float32) for _ in range(3))
        lib().orc_score(C.byref(c), _p(theta), _p(mu), _p(logmix), _p(sp), _p(costs), _p(actions), C.c_float(alpha), _p(sa),
                        _p(gl), _p(gp), _p(sc))
        return gl, gp, sc

    # -- a10 / a11 / IMQ
    def phi_k1(self, theta, score, variant=0, want_gram=False):
        theta, score = _f(theta), _f(score)
        N = theta.shape[0]
        phi = np.empty_like(theta)
        gram = np.empty((N, N), np.float32) if want_gram else None
        lib().orc_phi_k1(C.c_int(N), C.c_int(theta.size // N), _p(theta), _p(score), C.c_int(variant), _p(phi), _p(gram))
        return (phi, gram) if want_gram else phi

    def phi_imq(self, theta, score, ell):
        theta, score = _f(theta), _f(score)
        N = theta.shape[0]
        phi = np.empty_like(theta)
        lib().orc_phi_imq(C.c_int(N), C.c_int(theta.size // N), _p(theta), _p(score), C.c_float(ell), _p(phi))
        return phi

    def phi_k2(self, theta, score, indep=True, bw_scale=1.0, bandwidth=-1.0, minimum_bw=1e-5):
        c = self.c
        theta, score = _f(theta), _f(score)
        N = theta.shape[0]
        phi = np.empty_like(theta)
        h = np.empty(self.D if indep else c.H, np.float32)
        lib().orc_phi_k2(C.c_int(N), C.c_int(c.H), C.c_int(c.da), C.c_int(int(indep)), C.c_float(bw_scale), C.c_float(bandwidth),
                         C.c_float(minimum_bw), _p(theta), _p(score), _p(phi), _p(h))
        return phi, h

    @staticmethod
    def sgd(theta, phi, lr):
        theta, phi = _f(theta).copy(), _f(phi)
        lib().orc_sgd(C.c_int(theta.size), C.c_float(lr), _p(phi), _p(theta))
        return theta

    @staticmethod
    def adam(theta, phi, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8):
        """torch.optim.Adam step (svgd.py:115); returns (theta, m, v).  `step` is 1-based and restarts after every roll."""
        theta, phi, m, v = _f(theta).copy(), _f(phi), _f(m).copy(), _f(v).copy()
        lib().orc_adam(C.c_int(theta.size), C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps), C.c_int(step),
                       _p(phi), _p(theta), _p(m), _p(v))
        return theta, m, v

    # -- a12
    def forward(self, costs, theta, mu, mix_weights, sigma_p, alpha, lik=LIK_EXP_UTILITY, weighted_prior=False, roll=ROLL_REPEAT):
        c = self.c
        costs = _f(costs)
        theta, mu, mix = _f(theta).copy(), _f(mu).copy(), _f(mix_weights).copy()
        sp = _f(np.broadcast_to(sigma_p, (c.da,)))
        log_l, log_p, pw = (np.empty(c.N, np.float32) for _ in range(3))
        a_seq = np.empty((c.H, c.da), np.float32)
        istar = C.c_int(0)
        lib().orc_forward(C.byref(c), C.c_int(lik), C.c_float(alpha), _p(costs), _p(theta), _p(mu), _p(mix), _p(sp),
                          C.c_int(int(weighted_prior)), C.c_int(roll), _p(log_l), _p(log_p), _p(pw), C.byref(istar), _p(a_seq))
        return dict(log_l=log_l, log_p=log_p, p_weights=pw, i_star=istar.value, a_seq=a_seq, theta=theta, mu=mu, mix=mix)

    # -- a14
    def disco_step(self, a_mat, a_mix, strategy, steps, min_a, max_a, ext=None):
        c = self.c
        a_mat, a_mix = _f(a_mat).copy(), _f(a_mix)
        a_seq = np.zeros((c.H, c.da), np.float32)
        nxt = np.empty((steps, c.da), np.float32)
        sid = {"argmax": 0, "average": 1, "external": 2}[strategy]
        lo, hi = _f(np.broadcast_to(min_a, (c.da,))), _f(np.broadcast_to(max_a, (c.da,)))
        ext = None if ext is None else _f(ext)
        lib().orc_disco_step(C.c_int(c.N), C.c_int(c.H), C.c_int(c.da), C.c_int(sid), C.c_int(steps), _p(lo), _p(hi), _p(ext),
                             _p(a_mat), _p(a_mix), _p(a_seq), _p(nxt))
        return nxt, a_seq, a_mat

    def model_step(self, states, actions, params=None):
        c = self.c
        states, actions = _f(states).reshape(-1, c.ds), _f(actions).reshape(-1, c.da)
        params = None if params is None else _f(params).reshape(states.shape[0], -1)
        out = np.empty_like(states)
        lib().orc_model_step(C.byref(c), C.c_int(states.shape[0]), _p(states), _p(actions), C.c_int(actions.shape[0]), _p(params), _p(out))
        return out

    def get_collisions(self, xy):
        xy = _f(xy).reshape(-1, 2)
        out = np.empty(xy.shape[0], np.float32)
        lib().orc_get_collisions(C.byref(self.c), C.c_int(xy.shape[0]), _p(xy), _p(out))
        return out

    # -- a13
    def mpf_phi(self, x, prior_means, prior_bw, past_obs, past_action, obs, obs_std, log_space, bw):
        x, pm, po, pa, ob = _f(x), _f(prior_means), _f(past_obs), _f(past_action).reshape(-1), _f(obs)
        phi = np.empty_like(x)
        lib().orc_mpf_phi(C.byref(self.c), C.c_int(x.shape[0]), _p(x), _p(pm), C.c_float(prior_bw), _p(po), _p(pa), _p(ob),
                          C.c_float(obs_std), C.c_int(int(log_space)), C.c_float(bw), _p(phi))
        return phi

    def mpf_phi_v(self, x, prior_means, prior_bwv, past_obs, past_action, obs, obs_std, log_space, bw):
        """MPF.phi with a per-dimension prior bandwidth (the first prior of MPF(bw=None), mpf.py:31-32)."""
        x, pm, po, pa, ob, bv = _f(x), _f(prior_means), _f(past_obs), _f(past_action).reshape(-1), _f(obs), _f(prior_bwv)
        phi = np.empty_like(x)
        lib().orc_mpf_phi_v(C.byref(self.c), C.c_int(x.shape[0]), _p(x), _p(pm), _p(bv), _p(po), _p(pa), _p(ob),
                            C.c_float(obs_std), C.c_int(int(log_space)), C.c_float(bw), _p(phi))
        return phi

    def mpf_optimize_v(self, x, prior_means, prior_bwv, past_obs, past_action, obs, obs_std, log_space, bw, lr, n_steps):
        x, pm, po, pa, ob = _f(x).copy(), _f(prior_means).copy(), _f(past_obs), _f(past_action).reshape(-1), _f(obs)
        bv = _f(prior_bwv).copy()
        gn = np.empty(n_steps, np.float32)
        lib().orc_mpf_optimize_v(C.byref(self.c), C.c_int(x.shape[0]), _p(x), _p(pm), _p(bv), _p(po), _p(pa), _p(ob),
                                 C.c_float(obs_std), C.c_int(int(log_space)), C.c_float(bw), C.c_float(lr), C.c_int(n_steps), _p(gn))
        return x, pm, bv, gn

    @staticmethod
    def gmm_log_prob_v(x, means, bwv):
        x, means, bv = _f(x), _f(means), _f(bwv)
        out = np.empty(x.shape[0], np.float32)
        lib().orc_gmm_log_prob_v(C.c_int(x.shape[0]), C.c_int(means.shape[0]), C.c_int(x.shape[1]), _p(x), _p(means), _p(bv), _p(out))
        return out

    def mpf_optimize(self, x, prior_means, prior_bw, past_obs, past_action, obs, obs_std, log_space, bw, lr, n_steps):
        x, pm, po, pa, ob = _f(x).copy(), _f(prior_means).copy(), _f(past_obs), _f(past_action).reshape(-1), _f(obs)
        gn = np.empty(n_steps, np.float32)
        pbw = C.c_float(prior_bw)
        lib().orc_mpf_optimize(C.byref(self.c), C.c_int(x.shape[0]), _p(x), _p(pm), C.byref(pbw), _p(po), _p(pa), _p(ob),
                               C.c_float(obs_std), C.c_int(int(log_space)), C.c_float(bw), C.c_float(lr), C.c_int(n_steps), _p(gn))
        return x, pm, pbw.value, gn

    def mpf_optimize_adam(self, x, prior_means, prior_bw, past_obs, past_action, obs, obs_std, log_space, bw, lr, n_steps, m=None, v=None,
                          step=0, betas=(0.9, 0.999), eps=1e-8):
        """MPF.optimize with the class-default Adam optimiser; (m, v, step) carry the optimiser state between calls."""
        x, pm, po, pa, ob = _f(x).copy(), _f(prior_means).copy(), _f(past_obs), _f(past_action).reshape(-1), _f(obs)
        m = np.zeros_like(x) if m is None else _f(m).copy()
        v = np.zeros_like(x) if v is None else _f(v).copy()
        gn = np.empty(n_steps, np.float32)
        pbw, st = C.c_float(prior_bw), C.c_int(int(step))
        lib().orc_mpf_optimize_adam(C.byref(self.c), C.c_int(x.shape[0]), _p(x), _p(pm), C.byref(pbw), _p(po), _p(pa), _p(ob),
                                    C.c_float(obs_std), C.c_int(int(log_space)), C.c_float(bw), C.c_float(lr), C.c_float(betas[0]),
                                    C.c_float(betas[1]), C.c_float(eps), _p(m), _p(v), C.byref(st), C.c_int(n_steps), _p(gn))
        return x, pm, pbw.value, gn, m, v, st.value

    @staticmethod
    def gmm_log_prob(x, means, bw):
        x, means = _f(x), _f(means)
        out = np.empty(x.shape[0], np.float32)
        lib().orc_gmm_log_prob(C.c_int(x.shape[0]), C.c_int(means.shape[0]), C.c_int(x.shape[1]), _p(x), _p(means), C.c_float(bw), _p(out))
        return out

    # -- skid-steer family (f.4)
    @staticmethod
    def skid_rollout_cost(state, actions, params=None, uncertain_params=(), x_icr=0.2, wheel_radius=0.0625, axial_distance=0.475, dt=0.05,
                          lo=(-0.5, -0.5), hi=(0.5, 0.5), goal=(0, 0, 0, 0, 0), w_state=(1, 1, 1, 1, 1), w_term=(1, 1, 1, 1, 1), w_ctrl=(0, 0),
                          log_space=False, interleave=False, want_states=False):
        """SkidSteerRobot rollouts + the quadratic cost family: costs [S][N] (and states [M][S][N][H+1][5])."""
        actions = _f(actions)
        S, N, H = actions.shape[:3]
        names = ("x_icr", "wheel_radius", "axial_distance")
        cols = (C.c_int * 3)(*[list(uncertain_params).index(k) if k in uncertain_params else -1 for k in names])
        dflt = (C.c_double * 3)(x_icr, wheel_radius, axial_distance)
        pr = None if params is None else _f(params)
        M, P = (1, 0) if pr is None else pr.shape
        costs = np.empty((S, N), np.float32)
        st = np.empty((M, S, N, H + 1, 5), np.float32) if want_states else None
        lib().orc_skid_rollout_cost(C.c_int(N), C.c_int(S), C.c_int(M), C.c_int(H), C.c_int(P), _p(_f(state)), _p(actions), _p(pr), dflt, cols,
                                    C.c_int(int(log_space)), C.c_int(int(interleave)), C.c_double(dt), _p(_f(lo)), _p(_f(hi)), _p(_f(goal)),
                                    _p(_f(w_state)), _p(_f(w_term)), _p(_f(w_ctrl)), _p(costs), _p(st))
        return (costs, st) if want_states else costs

    # -- whole tick (cpu_baseline timing)
    def tick_k1(self, state, theta, mu, mix, sigma_p, sigma_a, eps, n_iters, alpha, lr, a_mat):
        c = self.c
        state, eps = _f(state).reshape(-1), _f(eps)
        theta, mu, mix, a_mat = _f(theta).copy(), _f(mu).copy(), _f(mix).copy(), _f(a_mat).copy()
        sp, sa = _f(np.broadcast_to(sigma_p, (c.da,))), _f(np.broadcast_to(sigma_a, (c.da,)))
        a_seq = np.empty((c.H, c.da), np.float32)
        pw = np.empty(c.N, np.float32)
        costs = np.empty((c.S, c.N), np.float32)
        lib().orc_tick_k1(C.byref(c), _p(state), _p(theta), _p(mu), _p(mix), _p(sp), _p(sa), _p(eps), C.c_int(n_iters),
                          C.c_float(alpha), C.c_float(lr), _p(a_mat), _p(a_seq), _p(pw), _p(costs))
        return dict(theta=theta, mu=mu, mix=mix, a_mat=a_mat, a_seq=a_seq, p_weights=pw, costs=costs)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    """OpenMP thread count of the following oracle calls; False when the oracle was built without OpenMP."""
    return bool(lib().orc_set_num_threads(C.c_int(int(n))))
