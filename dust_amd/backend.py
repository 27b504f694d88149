"""Thin object wrapper over the C ABI (one `Context` = one dust_ctx = one controller + its SVMPC state on one GPU).

All array arguments are numpy (or anything np.asarray accepts, torch CPU tensors included); results are numpy fp32.
No arithmetic of the hot path happens here - only shape checks and pointer plumbing.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _f(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = np.ascontiguousarray(a.reshape(shape))
    return a


def _fh(a, shape):
    """Bulk noise / action input: a float16 array stays binary16 (storage only, DUST_EPS_F16); anything else becomes fp32.
    Returns (array or None, flag)."""
    if a is None:
        return None, 0
    a = np.asarray(a)
    if a.dtype == np.float16:
        return np.ascontiguousarray(a.reshape(shape)), L.EPS_F16
    return _f(a, shape), 0


def _p(a):
    return None if a is None else a.ctypes.data_as(L.FP)


def _vp(a):
    return None if a is None else C.cast(a.ctypes.data_as(L.FP), L.VP)


def make_config(model="pendulum", N=1, S=1, M=1, H=1, uncertain_params=None, params_scalar_event=False,
                params_log_space=False, kernel="K1", likelihood="ExponentiatedUtility", optimizer="SGD", lr=1.0,
                alpha=1.0, temperature=None, ctrl_penalty=1.0, sigma_a=1.0, sigma_p=1.0, chol_a=None, a_pre=None,
                weighted_prior=False, roll_strategy="repeat", bw_scale=1.0, imq_ell=1.0, seed=0, device=0,
                shard_offset=0, shard_size=0, dt=None, g=9.8, mass=1.0, length=1.0, mass_0dim=False, w_cos=50.0, w_vel=1.0,
                max_speed=5.0, max_accel=10.0, can_crash=True, with_obstacle=True, cell_size=0.1,
                target=(9.0, 9.0, 0.0, 0.0), w_state=(0.5, 0.5, 0.25, 0.25), w_term=(1e3, 1e3, 0.1, 0.1), w_ctrl=(0.2, 0.2),
                w_obs=1e6, min_a=None, max_a=None, adam=(0.9, 0.999, 1e-8), sampling=None, control_type="acceleration",
                deterministic=True, noise_std=(0.0, 0.0), a_cov=None, p_cov=None, **_ignored):
    """a_cov / p_cov: FULL [da, da] action / prior-component covariances (MultiDISCO(a_cov=) disco.py:91-98; get_gmm svgd.py:84-89);
    they take precedence over sigma_a / sigma_p / chol_a / a_pre, which describe diagonal ones."""
    c = L.Config()
    c.abi_version = L.ABI_VERSION
    c.device = device
    pend = model == "pendulum"
    skid = model == "skid_steer"
    if model not in ("pendulum", "particle", "skid_steer"):
        raise ValueError("unknown model family %r" % (model,))
    c.model = L.MODEL_PENDULUM if pend else (L.MODEL_SKID_STEER if skid else L.MODEL_PARTICLE)
    c.cost = L.COST_PENDULUM_QUADCOS if pend else (L.COST_QUADRATIC if skid else L.COST_PARTICLE_DEFAULT)
    c.n_policies, c.n_samples, c.n_params, c.horizon = N, S, M, H
    if control_type not in ("acceleration", "velocity"):
        raise IOError('control_type "{}" not recognized'.format(control_type))  # particle.py:59-60
    vel = control_type == "velocity" and not pend and not skid
    c.dim_a, c.dim_s = (1, 2) if pend else ((2, 5) if skid else ((2, 2) if vel else (2, 4)))
    if not pend and not skid:  # Particle(control_type=, deterministic=, noise_std=) particle.py:13-31
        c.control_type = L.CONTROL_VELOCITY if vel else L.CONTROL_ACCELERATION
        c.ctrl_noise = int(not deterministic)
        ns = np.broadcast_to(np.asarray(noise_std, np.float32).reshape(-1), (2,))
        c.dyn_std[0], c.dyn_std[1] = float(ns[0]), float(ns[1])
    up = list(uncertain_params) if uncertain_params else []
    use_params = bool(up) if sampling is None else bool(sampling)
    c.dim_p = len(up) if use_params else 0
    c.shard_offset, c.shard_size = shard_offset, shard_size
    c.kernel = {"K1": L.KERNEL_K1_RBF, "K2": L.KERNEL_K2_IIDMP, "K2shared": L.KERNEL_K2_SHARED, "IMQ": L.KERNEL_IMQ}[kernel]
    c.likelihood = L.LIK_EXP_UTILITY if likelihood == "ExponentiatedUtility" else L.LIK_EXPECTED_COST
    c.optimizer = L.OPT_SGD if optimizer == "SGD" else L.OPT_ADAM
    c.roll_strategy = {"repeat": L.ROLL_REPEAT, "mean": L.ROLL_MEAN, "resample": L.ROLL_RESAMPLE}[roll_strategy]
    c.weighted_prior = int(weighted_prior)
    c.params_log_space = int(params_log_space)
    c.params_interleave = int(params_scalar_event)
    c.alpha = alpha
    c.temperature = (1.0 / alpha) if temperature is None else temperature
    c.a_reg = np.float32(c.temperature * (1.0 - ctrl_penalty))
    c.lr = lr
    c.adam_beta1, c.adam_beta2, c.adam_eps = adam
    sa = np.broadcast_to(np.asarray(sigma_a, np.float32), (c.dim_a,))
    sp = np.broadcast_to(np.asarray(sigma_p, np.float32), (c.dim_a,))
    ch = sa if chol_a is None else np.broadcast_to(np.asarray(chol_a, np.float32), (c.dim_a,))
    ap = (1.0 / (sa.astype(np.float32) ** 2)) if a_pre is None else np.broadcast_to(np.asarray(a_pre, np.float32), (c.dim_a,))
    for d in range(c.dim_a):
        c.sigma_a[d], c.sigma_p[d], c.chol_a[d], c.a_pre[d] = sa[d], sp[d], ch[d], ap[d]
    if a_cov is not None or p_cov is not None:
        import torch  # (the factorisations in fp32, as the reference's torch.linalg.cholesky / torch.inverse produce them)

        ac = torch.diag(torch.as_tensor(sa.astype(np.float32)) ** 2) if a_cov is None else torch.as_tensor(np.asarray(a_cov, np.float32))
        pc = torch.diag(torch.as_tensor(sp.astype(np.float32)) ** 2) if p_cov is None else torch.as_tensor(np.asarray(p_cov, np.float32))
        if tuple(ac.shape) != (c.dim_a, c.dim_a) or tuple(pc.shape) != (c.dim_a, c.dim_a):
            raise ValueError("a_cov / p_cov must be [%d, %d] matrices" % (c.dim_a, c.dim_a))
        la, lp, pre = torch.linalg.cholesky(ac), torch.linalg.cholesky(pc), torch.inverse(ac)
        for d in range(c.dim_a):
            c.chol_a[d], c.a_pre[d] = float(la[d, d]), float(pre[d, d])
            c.sigma_a[d], c.sigma_p[d] = float(ac[d, d].sqrt()), float(pc[d, d].sqrt())
        off = c.dim_a == 2 and (float(ac[0, 1]) != 0.0 or float(ac[1, 0]) != 0.0 or float(pc[0, 1]) != 0.0 or float(pc[1, 0]) != 0.0)
        if off:
            c.full_cov = 1
            c.chol_a_off, c.a_pre_off = float(la[1, 0]), float(pre[0, 1])
            c.chol_p[0], c.chol_p[1], c.chol_p[2] = float(lp[0, 0]), float(lp[1, 0]), float(lp[1, 1])
        elif c.dim_a > 2 or not torch.equal(ac, torch.diag(torch.diag(ac))) or not torch.equal(pc, torch.diag(torch.diag(pc))):
            raise NotImplementedError("full covariances are implemented for dim_a = 2")
    c.bw_scale, c.imq_ell = bw_scale, imq_ell
    lo = np.broadcast_to(np.asarray((-2.0 if pend else (-0.5 if skid else -max_accel)) if min_a is None else min_a, np.float32), (c.dim_a,))
    hi = np.broadcast_to(np.asarray((2.0 if pend else (0.5 if skid else max_accel)) if max_a is None else max_a, np.float32), (c.dim_a,))
    for d in range(c.dim_a):
        c.min_a[d], c.max_a[d] = lo[d], hi[d]
    c.seed = seed
    if skid and dt is None:
        raise ValueError("SkidSteerRobot(delta_t=...) has no default: pass dt")
    c.dt = (0.05 if pend else 0.015) if dt is None else dt

    def par(name, value):
        if use_params and name in up:
            return L.Param(L.PARAM_SAMPLED, up.index(name), float(value))
        return L.Param(L.PARAM_TENSOR0D if (mass_0dim and name == "mass") else L.PARAM_PYFLOAT, 0, float(value))

    c.g, c.mass, c.length = par("g", g), par("mass", mass), par("length", length)
    c.max_torque, c.max_speed_pend, c.w_cos, c.w_vel = 2.0, 8.0, w_cos, w_vel
    c.max_speed, c.max_accel = max_speed, max_accel
    c.can_crash, c.with_obstacle = int(can_crash), int(with_obstacle and not pend and not skid)
    c.cell_size = cell_size
    pad4 = lambda v: (list(np.asarray(v, np.float32).reshape(-1)) + [0.0] * 4)[:4]  # (velocity control: two-entry target / weights)
    c.target[:] = pad4(target)
    c.w_state[:] = pad4(w_state)
    c.w_term[:] = pad4(w_term)
    c.w_ctrl[:] = list(w_ctrl)
    c.w_obs = w_obs
    return c


class Context:
    def __init__(self, cfg=None, grid=None, _handle=None, **kw):
        self._h = None
        lib = L.load()
        k2_bw, k2_min = kw.pop("k2_bandwidth", None), kw.pop("k2_minimum_bw", 1e-5)
        skid_kw = {k: kw.pop(k) for k in ("x_icr", "wheel_radius", "axial_distance", "goal", "w_quad_state", "w_quad_term", "w_quad_ctrl")
                   if k in kw}
        if _handle is not None:
            self._h = _handle
        else:
            self.cfg = cfg if cfg is not None else make_config(**kw)
            h = L.VP()
            L.check(lib.dust_create(C.byref(self.cfg), C.byref(h)))
            self._h = h
        got = L.Config()
        L.check(lib.dust_get_config(self._h, C.byref(got)))
        self.cfg = got
        self.N, self.S, self.M, self.H = got.n_policies, got.n_samples, got.n_params, got.horizon
        self.da, self.ds, self.P = got.dim_a, got.dim_s, got.dim_p
        self.D = self.H * self.da
        # per-tick buffers of the control loop with their ctypes pointers made ONCE: `array.ctypes.data_as(...)` costs 2.8 us per call, and
        # a closed-loop tick needs three of them - 8 us of its 19 us beyond the kernel (tools/closed_loop_probe.py)
        self._tick_state = np.zeros(self.ds, np.float32)
        self._tick_aseq = np.zeros((self.H, self.da), np.float32)
        self._tick_pw = np.zeros(self.N, np.float32)
        self._tick_ptrs = (_p(self._tick_state), _p(self._tick_aseq), _p(self._tick_pw))
        self._tick_fn = lib.dust_svmpc_tick
        if grid is not None:
            self.set_grid(grid)
        if _handle is None and got.model == L.MODEL_SKID_STEER:
            self.set_skid_steer(uncertain_params=kw.get("uncertain_params"), sampling=kw.get("sampling"), **skid_kw)
        # iid_mp(RBF(bandwidth >= 0)): fixed bandwidth instead of the median trick; RBF(minimum_bw=): the clamp of either (base_kernels.py:44-92)
        if (k2_bw is not None and float(k2_bw) >= 0) or (_handle is None and float(k2_min) != 1e-5 and got.kernel in (L.KERNEL_K2_IIDMP, L.KERNEL_K2_SHARED)):
            L.check(lib.dust_set_k2_bandwidth(self._h, float(-1.0 if k2_bw is None else k2_bw), float(k2_min)))

    # ---- lifecycle
    # ---- C-side multi-GPU tick (include/dust_amd.h dust_comm_*): one RCCL communicator per sharded context
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId on the calling rank: 128 opaque bytes to hand to every rank (e.g. torch.distributed.broadcast_object_list)."""
        buf = C.create_string_buffer(128)
        L.check(L.load().dust_comm_unique_id(C.cast(buf, L.VP)))
        return buf.raw

    def comm_validate(self, rank, world):
        """The local (non-collective) checks of comm_init; raises on failure.  Call it on every rank and let the ranks agree before comm_init."""
        L.check(L.load().dust_comm_validate(self._h, int(rank), int(world)))

    def comm_init(self, unique_id, rank, world):
        """ncclCommInitRank (collective over the ranks).  Afterwards svmpc_tick / svmpc_optimize / svmpc_forward run the sharded tick."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(L.load().dust_comm_init(self._h, C.cast(buf, L.VP), int(rank), int(world)))

    def comm_probe(self, n_steps, reps=50):
        """Microseconds per tick of the sharded tick's all-gathers alone (collective over all ranks)."""
        us = C.c_double(0.0)
        L.check(L.load().dust_comm_probe(self._h, int(n_steps), int(reps), C.byref(us)))
        return float(us.value)

    def dual_tick(self, mpf, state, action_prev, n_steps, mpf_steps=20, mpf_bw=None, seed=0, want_outputs=True):
        """One control period of the dual loop in one C call (dust_dual_tick): filter update for (action_prev, state) - skipped when
        action_prev is None -, the controller's dynamics samples drawn from the filter's refreshed prior on the device, the control
        tick.  mpf_bw None: Silverman's rule of the filter's particles, on the device.  Returns (a_seq, p_weights, bw_used)."""
        st = _f(state, (self.ds,))
        ap = None if action_prev is None else _f(action_prev).reshape(-1)
        a_seq = np.empty((self.H, self.da), np.float32) if want_outputs else None
        pw = np.empty(self.N, np.float32) if want_outputs else None
        bw = C.c_float(0.0)
        L.check(L.load().dust_dual_tick(self._h, mpf._h, _p(st), _p(ap), int(n_steps), int(mpf_steps), float(-1.0 if mpf_bw is None else mpf_bw),
                                        int(seed), _p(a_seq), _p(pw), C.cast(C.byref(bw), L.FP)))
        return a_seq, pw, float(bw.value)

    def comm_peer_gather(self, on=True):
        """The tick's all-gathers as direct peer stores through IPC-mapped buffers (collective: every rank calls it, or none)."""
        L.check(L.load().dust_comm_peer_gather(self._h, 1 if on else 0))

    def close(self):
        if self._h is not None:
            L.load().dust_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self):
        h = L.VP()
        L.check(L.load().dust_clone(self._h, C.byref(h)))
        return Context(_handle=h)

    def __deepcopy__(self, memo):
        return self.clone()

    def sync(self):
        L.check(L.load().dust_sync(self._h))

    def tick_stats(self):
        """Sticky counts of the device path that served optimize / tick calls: owner-computes one-launch ticks, tiled one-launch
        ticks, and ticks replayed on the launch-per-iteration path because the device was shared."""
        n = (C.c_longlong * 4)()
        L.check(L.load().dust_tick_stats(self._h, n))
        return dict(tick2=int(n[0]), tick1=int(n[1]), served=int(n[2]), replayed=int(n[3]))

    def serve_start(self, n_steps, wait_us=2000.0):
        """Closed-loop serving (dust_svmpc_serve_start): svmpc_tick(state, n_steps) calls receive their outputs through pinned host
        memory and launch the next tick ahead of its plant state (it waits at most wait_us for it).  Same results, bit for bit."""
        L.check(L.load().dust_svmpc_serve_start(self._h, int(n_steps), float(wait_us)))

    def serve_stop(self):
        L.check(L.load().dust_svmpc_serve_stop(self._h))

    def set_grid(self, grid, off=None):
        g = _f(grid)
        nx, ny = g.shape
        ox, oy = (int(nx / 2), int(ny / 2)) if off is None else off
        L.check(L.load().dust_set_grid(self._h, _p(g), nx, ny, float(ox), float(oy)))

    def set_skid_steer(self, x_icr=0.2, wheel_radius=0.0625, axial_distance=0.475, goal=(0, 0, 0, 0, 0), w_quad_state=(1, 1, 1, 1, 1),
                       w_quad_term=(1, 1, 1, 1, 1), w_quad_ctrl=(0, 0), uncertain_params=None, sampling=None):
        """SkidSteerRobot parameters (skid_steer_robot.py:19-45; `uncertain_params` name the sampled ones, in column order) and the
        quadratic cost family (dust_amd.costs.QuadraticCost).  Wheel-speed bounds are the context's min_a / max_a."""
        up = list(uncertain_params) if uncertain_params else []
        use = bool(up) if sampling is None else bool(sampling)
        g = L.SkidConfig()

        def par(name, value):
            if use and name in up:
                return L.Param(L.PARAM_SAMPLED, up.index(name), float(value))
            return L.Param(L.PARAM_PYFLOAT, 0, float(value))

        g.x_icr, g.wheel_radius, g.axial_distance = par("x_icr", x_icr), par("wheel_radius", wheel_radius), par("axial_distance", axial_distance)
        for d in range(2):
            g.min_wheel_speed[d], g.max_wheel_speed[d] = self.cfg.min_a[d], self.cfg.max_a[d]
        g.goal[:] = [float(v) for v in goal]
        g.w_state[:] = [float(v) for v in w_quad_state]
        g.w_term[:] = [float(v) for v in w_quad_term]
        g.w_ctrl[:] = [float(v) for v in w_quad_ctrl]
        L.check(L.load().dust_set_skid_steer(self._h, C.byref(g)))

    def set_ctrl_noise(self, z):
        """Recorded control-channel noise for the next rollouts of a Particle(deterministic=False) context (particle.py:145-148):
        z [n_sets][H][M*S*N][da] standard-normal draws in the reference's own order (one randn_like per model.step call); None: back
        to the device Philox stream."""
        if z is None:
            L.check(L.load().dust_set_ctrl_noise(self._h, None, 0))
            return
        z = _f(z, (-1, self.H, self.M * self.S * self.N, self.da))
        L.check(L.load().dust_set_ctrl_noise(self._h, _p(z), int(z.shape[0])))

    def set_param_weights(self, w):
        """Unscented-transform weights of the n_params dynamics samples (None: plain mean)."""
        L.check(L.load().dust_set_param_weights(self._h, _p(None if w is None else _f(w, (self.M,)))))

    def set_model_param(self, name, value, kind=-1):
        L.check(L.load().dust_set_model_param(self._h, name.encode(), float(value), kind))

    # ---- state
    def _get(self, fn, shape):
        out = np.empty(shape, np.float32)
        L.check(fn(self._h, _p(out)))
        return out

    def set_theta(self, theta):
        L.check(L.load().dust_set_theta(self._h, _p(_f(theta, (self.N, self.H, self.da)))))

    def get_theta(self):
        return self._get(L.load().dust_get_theta, (self.N, self.H, self.da))

    def set_prior(self, means, weights=None):
        w = None if weights is None else _f(weights, (self.N,))
        L.check(L.load().dust_set_prior(self._h, _p(_f(means, (self.N, self.H, self.da))), _p(w)))

    def get_prior(self):
        means = np.empty((self.N, self.H, self.da), np.float32)
        probs = np.empty(self.N, np.float32)
        L.check(L.load().dust_get_prior(self._h, _p(means), _p(probs)))
        return means, probs

    def set_a_mat(self, a):
        L.check(L.load().dust_set_a_mat(self._h, _p(_f(a, (self.N, self.H, self.da)))))

    def get_a_mat(self):
        return self._get(L.load().dust_get_a_mat, (self.N, self.H, self.da))

    def get_a_mix(self):
        return self._get(L.load().dust_get_a_mix, (self.N,))

    def set_a_seq(self, a):
        L.check(L.load().dust_set_a_seq(self._h, _p(_f(a, (self.H, self.da)))))

    def get_a_seq(self):
        return self._get(L.load().dust_get_a_seq, (self.H, self.da))

    # ---- a1-a6
    def _params(self, params, sets=1):
        if self.P == 0:
            return None
        if params is None:
            raise ValueError("params_sampling is on: pass the sampled dynamics parameters")
        return _f(params, (sets, self.M, self.P))

    def disco_forward(self, state, actions=None, params=None, want_states=False, want_actions=False, want_omega=True,
                      around_a_mat=False, store_f16=False):
        """`actions` may be a float16 array (binary16 storage, fp32 arithmetic); `store_f16`: states / actions come back as float16."""
        st = _f(state, (self.ds,))
        act, fl = _fh(actions, (self.S, self.N, self.H, self.da))
        pr = self._params(params)
        odt = np.float16 if store_f16 else np.float32
        costs = np.empty((self.S, self.N), np.float32)
        states = np.empty((self.M, self.S, self.N, self.H + 1, self.ds), odt) if want_states else None
        aout = np.empty((self.S, self.N, self.H, self.da), odt) if want_actions else None
        omega = np.empty((self.S, self.N), np.float32) if want_omega else None
        flags = (L.EPS_AROUND_A_MAT if around_a_mat else 0) | fl | (L.STORE_F16 if store_f16 else 0)
        L.check(L.load().dust_disco_forward(self._h, _p(st), _vp(act), _p(pr), flags, _p(costs), _p(states), _p(aout), _p(omega)))
        return costs, states, (aout if aout is not None else act), omega

    def disco_step(self, strategy="argmax", steps=1, ext_actions=None):
        sid = {"argmax": L.STEP_ARGMAX, "average": L.STEP_AVERAGE, "external": L.STEP_EXTERNAL}.get(strategy, -1)
        ext = None if ext_actions is None else _f(ext_actions, (self.H, self.da))
        out = np.empty((steps, self.da), np.float32)
        L.check(L.load().dust_disco_step(self._h, sid, steps, _p(ext), _p(out)))
        return out

    def likelihood_sample(self, state, eps=None, params=None, want_actions=False, store_states=False, store_f16=False):
        st = _f(state, (self.ds,))
        e, fl = _fh(eps, (self.S, self.N, self.H, self.da))
        pr = self._params(params)
        costs = np.empty((self.S, self.N), np.float32)
        aout = np.empty((self.S, self.N, self.H, self.da), np.float16 if store_f16 else np.float32) if want_actions else None
        flags = (L.STORE_STATES if store_states else 0) | fl | (L.STORE_F16 if store_f16 else 0)
        L.check(L.load().dust_likelihood_sample(self._h, _p(st), _vp(e), _p(pr), flags, _p(costs), _p(aout)))
        return (costs, aout) if want_actions else costs

    def likelihood_log_prob(self):
        return self._get(L.load().dust_likelihood_log_prob, (self.N,))

    # ---- a7-a12
    def svmpc_phi(self, costs=None, actions=None):
        c = None if costs is None else _f(costs, (self.S, self.N))
        a = None if actions is None else _f(actions, (self.S, self.N, self.H, self.da))
        phi, gl, gp = (np.empty((self.N, self.H, self.da), np.float32) for _ in range(3))
        L.check(L.load().dust_svmpc_phi(self._h, _p(c), _p(a), _p(phi), _p(gl), _p(gp)))
        return phi, gl, gp

    def svmpc_optimize(self, state, n_steps, eps=None, params=None):
        st = _f(state, (self.ds,))
        e, fl = _fh(eps, (n_steps, self.S, self.N, self.H, self.da))
        pr = self._params(params, n_steps)
        L.check(L.load().dust_svmpc_optimize(self._h, _p(st), n_steps, _vp(e), _p(pr), fl))

    def svmpc_optimize_dev(self, state, n_steps, eps_dev_ptr, params=None):
        st = _f(state, (self.ds,))
        pr = self._params(params, n_steps)
        L.check(L.load().dust_svmpc_optimize(self._h, _p(st), n_steps, L.VP(eps_dev_ptr) if eps_dev_ptr else None, _p(pr),
                                             L.PTR_DEVICE))

    def svmpc_forward(self):
        a_seq = np.empty((self.H, self.da), np.float32)
        pw = np.empty(self.N, np.float32)
        L.check(L.load().dust_svmpc_forward(self._h, _p(a_seq), _p(pw)))
        return a_seq, pw

    def likelihood_sample_at(self, state, theta, eps=None, params=None, want_actions=False):
        """CostLikelihood.sample(theta, ...) for a theta that is not the optimiser's: the context's particles / Adam state stay."""
        st = _f(state, (self.ds,))
        th = _f(theta, (self.N, self.H, self.da))
        e, fl = _fh(eps, (self.S, self.N, self.H, self.da))
        pr = self._params(params)
        costs = np.empty((self.S, self.N), np.float32)
        aout = np.empty((self.S, self.N, self.H, self.da), np.float32) if want_actions else None
        L.check(L.load().dust_likelihood_sample_at(self._h, _p(st), _p(th), _vp(e), _p(pr), fl, _p(costs), _p(aout)))
        return (costs, aout) if want_actions else costs

    def svmpc_get_weights(self):
        pw = np.empty(self.N, np.float32)
        L.check(L.load().dust_svmpc_get_weights(self._h, _p(pw)))
        return pw

    def svmpc_roll(self, steps=-1, strategy="repeat", last_row=None):
        st = {"repeat": L.ROLL_REPEAT, "mean": L.ROLL_MEAN, "resample": L.ROLL_RESAMPLE}.get(strategy)
        if st is None:
            raise ValueError("{} is an invalid roll strategy.".format(strategy))
        lr = None if last_row is None else _f(last_row, (self.N, self.da))
        L.check(L.load().dust_svmpc_roll(self._h, int(steps), st, _p(lr)))

    def svmpc_update_prior(self, weights=None):
        w = None if weights is None else _f(weights, (self.N,))
        L.check(L.load().dust_svmpc_update_prior(self._h, _p(w)))

    def svmpc_forward_ex(self, steps=-1, resample_last_row=None):
        a_seq, pw = np.empty((self.H, self.da), np.float32), np.empty(self.N, np.float32)
        lr = None if resample_last_row is None else _f(resample_last_row, (self.N, self.da))
        L.check(L.load().dust_svmpc_forward_ex(self._h, int(steps), _p(lr), _p(a_seq), _p(pw)))
        return a_seq, pw

    def svmpc_tick(self, state, n_steps, eps=None, params=None, eps_dev_ptr=None, want_outputs=True):
        if eps is None and params is None and not eps_dev_ptr and self.P == 0:
            # the control loop's call (device noise, nominal dynamics): cached buffers and pointers, the outputs handed back as copies
            self._tick_state[:] = np.asarray(state, np.float32).reshape(self.ds)
            ps, pa, pw_ = self._tick_ptrs
            if want_outputs == "action":  # the chosen sequence only (what a control loop sends to its plant): p_weights are not fetched
                L.check(self._tick_fn(self._h, ps, n_steps, None, None, 0, pa, None))
                return self._tick_aseq.copy(), None
            if want_outputs:
                L.check(self._tick_fn(self._h, ps, n_steps, None, None, 0, pa, pw_))
                return self._tick_aseq.copy(), self._tick_pw.copy()
            L.check(self._tick_fn(self._h, ps, n_steps, None, None, 0, None, None))
            return None, None
        st = _f(state, (self.ds,))
        pr = self._params(params, n_steps)
        a_seq = np.empty((self.H, self.da), np.float32) if want_outputs else None
        pw = np.empty(self.N, np.float32) if want_outputs and want_outputs != "action" else None
        if eps_dev_ptr:
            e, flags = L.VP(eps_dev_ptr), L.PTR_DEVICE
        else:
            e, flags = _fh(eps, (n_steps, self.S, self.N, self.H, self.da))
            e = _vp(e)
        L.check(L.load().dust_svmpc_tick(self._h, _p(st), n_steps, e, _p(pr), flags, _p(a_seq), _p(pw)))
        return a_seq, pw

    def get_costs(self):
        return self._get(L.load().dust_get_costs, (self.S, self.N))

    def get_states_rows(self, rows, f16=False):
        """Trajectories [len(rows)][H+1][ds] of the states the last stored-states sample left on the device; rows = (m*S + s)*N + n."""
        r = np.ascontiguousarray(np.asarray(rows, np.int64).reshape(-1))
        out = np.empty((r.size, self.H + 1, self.ds), np.float16 if f16 else np.float32)
        L.check(L.load().dust_get_states_rows(self._h, r.ctypes.data_as(C.POINTER(C.c_longlong)), int(r.size), C.c_void_p(out.ctypes.data)))
        return out

    def get_score(self):
        return self._get(L.load().dust_get_score, (self.N, self.H, self.da))

    def get_phi(self):
        return self._get(L.load().dust_get_phi, (self.N, self.H, self.da))

    def get_score_parts(self):
        """(grad_lik, grad_pri) of the last SVGD iteration (svmpc.py:38-53)."""
        gl, gp = (np.empty((self.N, self.H, self.da), np.float32) for _ in range(2))
        L.check(L.load().dust_get_score_parts(self._h, _p(gl), _p(gp)))
        return gl, gp

    def get_log_weights(self):
        ll, lp = np.empty(self.N, np.float32), np.empty(self.N, np.float32)
        L.check(L.load().dust_get_log_weights(self._h, _p(ll), _p(lp)))
        return ll, lp

    def get_bandwidths(self):
        G = self.H if self.cfg.kernel == L.KERNEL_K2_SHARED else self.D
        return self._get(L.load().dust_get_bandwidths, (G,))

    # ---- timing
    def profile(self, on=True):
        L.check(L.load().dust_profile_enable(self._h, int(on)))
        L.check(L.load().dust_profile_reset(self._h))

    def profile_get(self):
        out = {}
        for k in range(L.K_COUNT):
            ms, n = C.c_double(0), C.c_int64(0)
            L.check(L.load().dust_profile_get(self._h, k, C.byref(ms), C.byref(n)))
            if n.value:
                out[L.load().dust_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def profile_rollout(self, state, eps_dev, n_slices, reps, f16=False, store_states=False, store_f16=False):
        """Average launch-to-launch time (ms) of the standalone rollout kernel over device-resident eps slices
        (store_states: the HBM-bound form that also writes states [M][S][N][H+1][ds])."""
        ms = C.c_double(0)
        fl = (L.EPS_F16 if f16 else 0) | (L.STORE_STATES if store_states else 0) | (L.STORE_F16 if store_f16 else 0)
        L.check(L.load().dust_profile_rollout(self._h, _p(np.ascontiguousarray(state, np.float32)), C.c_void_p(eps_dev), int(n_slices),
                                              int(reps), fl, C.byref(ms)))
        return ms.value

    def rollout_bytes(self, store_states=False, eps_f16=False, store_f16=False):
        b = C.c_double(0)
        fl = (L.STORE_STATES if store_states else 0) | (L.EPS_F16 if eps_f16 else 0) | (L.STORE_F16 if store_f16 else 0)
        L.check(L.load().dust_rollout_algorithmic_bytes(self._h, fl, C.byref(b)))
        return b.value

    def device_noise(self, n_values, seed=1, f16=False):
        p = L.VP()
        L.check(L.load().dust_device_noise_alloc(self._h, n_values, seed, L.EPS_F16 if f16 else 0, C.byref(p)))
        return p.value

    def device_free(self, ptr):
        L.check(L.load().dust_device_free(self._h, L.VP(ptr)))


class MpfContext:
    def __init__(self, init_particles, initial_obs, model="pendulum", uncertain_params=("length", "mass"), log_space=False,
                 obs_std=0.1, lr=1e-3, bw_scale=1.0, init_bw=0.1, device=0, grid=None, _handle=None, optimizer="SGD", betas=(0.9, 0.999),
                 eps=1e-8, **model_kw):
        lib = L.load()
        x = _f(init_particles)
        self.Mp, self.P = x.shape
        if _handle is not None:
            self._h = _handle
            return
        c = L.MpfConfig()
        c.abi_version, c.device, c.n_particles, c.dim_p = L.ABI_VERSION, device, self.Mp, self.P
        c.model_cfg = make_config(model=model, uncertain_params=uncertain_params, params_log_space=log_space, device=device, **model_kw)
        c.dim_s, c.dim_a, c.model = c.model_cfg.dim_s, c.model_cfg.dim_a, c.model_cfg.model
        c.log_space, c.obs_std, c.lr, c.bw_scale, c.init_bw = int(log_space), obs_std, lr, bw_scale, init_bw
        self.ds, self.da = c.dim_s, c.dim_a
        h = L.VP()
        L.check(lib.dust_mpf_create(C.byref(c), _p(x), _p(_f(initial_obs, (c.dim_s,))), C.byref(h)))
        self._h = h
        if grid is not None:
            g = _f(grid)
            L.check(lib.dust_mpf_set_grid(self._h, _p(g), g.shape[0], g.shape[1], float(int(g.shape[0] / 2)), float(int(g.shape[1] / 2))))
        if optimizer not in ("SGD", "Adam"):
            raise NotImplementedError("MPF optimiser %r: the device filter implements SGD and Adam" % (optimizer,))
        if optimizer == "Adam":  # the reference's class default (svgd.py:115); its state persists across optimize() calls
            L.check(lib.dust_mpf_set_optimizer(self._h, L.OPT_ADAM, float(betas[0]), float(betas[1]), float(eps)))

    def close(self):
        if getattr(self, "_h", None) is not None:
            L.load().dust_mpf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def silverman(self):
        """silvermans_rule of the pooled particles times bw_scale, on the device (dust_mpf_silverman)."""
        bw = C.c_float(0.0)
        L.check(L.load().dust_mpf_silverman(self._h, C.cast(C.byref(bw), L.FP)))
        return float(bw.value)

    def clone(self):
        h = L.VP()
        L.check(L.load().dust_mpf_clone(self._h, C.byref(h)))
        m = MpfContext.__new__(MpfContext)
        m._h, m.Mp, m.P, m.ds, m.da = h, self.Mp, self.P, self.ds, self.da
        return m

    def __deepcopy__(self, memo):
        return self.clone()

    def condition(self, action, new_obs):
        a = None if action is None else _f(action).reshape(-1)
        L.check(L.load().dust_mpf_condition(self._h, _p(a), _p(_f(new_obs, (self.ds,)))))

    def set_ctrl_noise(self, z):
        """Recorded control-noise draws [n][da] for the next phi / optimize steps (one per SVGD step; likelihoods.py:30-46 ->
        particle.py:145-148); None: the library's own generator."""
        if z is None:
            L.check(L.load().dust_mpf_set_ctrl_noise(self._h, None, 0))
            return
        z = _f(z, (-1, self.da))
        L.check(L.load().dust_mpf_set_ctrl_noise(self._h, _p(z), int(z.shape[0])))

    def phi(self, bw):
        out = np.empty((self.Mp, self.P), np.float32)
        L.check(L.load().dust_mpf_phi(self._h, float(bw), _p(out)))
        return out

    def optimize(self, action, new_obs, bw, n_steps):
        a = None if action is None else _f(action).reshape(-1)
        o = None if new_obs is None else _f(new_obs, (self.ds,))
        gn = np.empty(max(n_steps, 1), np.float32)
        L.check(L.load().dust_mpf_optimize(self._h, _p(a), _p(o), float(bw), n_steps, _p(gn)))
        return gn[:n_steps]

    def get_particles(self):
        out = np.empty((self.Mp, self.P), np.float32)
        L.check(L.load().dust_mpf_get_particles(self._h, _p(out)))
        return out

    def set_particles(self, x):
        L.check(L.load().dust_mpf_set_particles(self._h, _p(_f(x, (self.Mp, self.P)))))

    def get_prior(self):
        means = np.empty((self.Mp, self.P), np.float32)
        bw = C.c_float(0)
        L.check(L.load().dust_mpf_get_prior(self._h, _p(means), C.cast(C.byref(bw), L.FP)))
        return means, bw.value

    def set_prior_bw(self, bw):
        """Per-dimension bandwidths of the current prior (MPF(bw=None): bw_silverman of the particle columns, mpf.py:31-36)."""
        b = np.ascontiguousarray(np.asarray(bw, np.float32).reshape(-1))
        L.check(L.load().dust_mpf_set_prior_bw(self._h, _p(b), int(b.size)))

    def get_prior_bw(self):
        out = np.empty(self.P, np.float32)
        L.check(L.load().dust_mpf_get_prior_bw(self._h, _p(out)))
        return out

    def stats(self):
        """{'grid': optimize() calls served by the multi-workgroup kernel, 'fallback': those re-run by the single-workgroup one}."""
        import ctypes as C
        n = (C.c_longlong * 2)()
        L.check(L.load().dust_mpf_stats(self._h, n))
        return {"grid": int(n[0]), "fallback": int(n[1])}

    def prior_sample(self, n, seed=0):
        out = np.empty((n, self.P), np.float32)
        L.check(L.load().dust_mpf_prior_sample(self._h, n, seed, _p(out)))
        return out

    def prior_log_prob(self, x):
        x = _f(x, (-1, self.P))
        out = np.empty(x.shape[0], np.float32)
        L.check(L.load().dust_mpf_prior_log_prob(self._h, x.shape[0], _p(x), _p(out)))
        return out
