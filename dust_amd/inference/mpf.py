"""`MPF` with the reference's constructor and `optimize` (dust/inference/mpf.py:13-86).  `.prior` offers what the controller
needs from a torch distribution (`sample`, `log_prob`, `event_shape`, `mean`), backed by the device particles."""
import numpy as np
import torch

from ..backend import MpfContext
from .svgd import bw_silverman


def silvermans_rule(data):
    """KDEpy 1.1.0 `bw_selection.silvermans_rule` restated (third-party, PARITY UNPINNED - SURVEY 8c):
    min(std(ddof=1), IQR/1.349) * (3n/4)^(-1/5)."""
    data = np.asarray(data, dtype=np.float64).reshape(-1)
    n = data.size
    if n == 1:
        return 1.0
    std = np.std(data, ddof=1)
    q75, q25 = np.percentile(data, [75, 25])
    iqr = (q75 - q25) / 1.3489795003921634
    sigma = min(std, iqr)
    if not sigma > 0:
        sigma = max(std, iqr)
    return float(sigma * (n * 3 / 4.0) ** (-1 / 5)) if sigma > 0 else 1.0


class _DevicePrior:
    def __init__(self, mpf):
        self._mpf = mpf
        self._seed = 0

    @property
    def event_shape(self):
        return torch.Size([self._mpf._dev.P])

    @property
    def mean(self):
        return torch.from_numpy(self._mpf._dev.get_prior()[0]).mean(0)

    def sample(self, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        self._seed += 1
        out = torch.from_numpy(self._mpf._dev.prior_sample(n, self._seed))
        return out.reshape(*shape, -1) if len(shape) else out.reshape(-1)

    def log_prob(self, x):
        x = torch.as_tensor(x, dtype=torch.float)
        return torch.from_numpy(self._mpf._dev.prior_log_prob(x.reshape(-1, self._mpf._dev.P).numpy())).reshape(x.shape[:-1])


class MPF:
    def __init__(self, init_particles, likelihood, bw=None, bw_scale=1.0, optimizer_class=torch.optim.Adam, n_steps=100, **opt_args):
        """optimizer_class: torch.optim.Adam (the reference's class default, svgd.py:115) or torch.optim.SGD (what the demos pass);
        as in the reference the optimiser is built once, so Adam's moments persist across optimize() calls (mpf.py:24)."""
        init_particles = torch.as_tensor(init_particles, dtype=torch.float)
        assert init_particles.ndim == 2, "Particles must be two dimension with batch on dim 0."
        if optimizer_class is torch.optim.SGD:
            opt_kw, allowed = dict(optimizer="SGD"), {"lr"}
        elif optimizer_class is torch.optim.Adam:
            opt_kw = dict(optimizer="Adam", betas=tuple(opt_args.get("betas", (0.9, 0.999))), eps=float(opt_args.get("eps", 1e-8)))
            allowed = {"lr", "betas", "eps"}
        else:
            raise NotImplementedError("MPF on the device implements torch.optim.SGD and torch.optim.Adam, not %r" % (optimizer_class,))
        extra = set(opt_args) - allowed
        if extra:
            raise NotImplementedError("optimiser options %s are not implemented on the device" % sorted(extra))
        self.likelihood, self.bw_scale = likelihood, bw_scale
        bw_vec = None
        if bw is None:  # mpf.py:31-32: a scalar (IQR branch of _select_sigma) or one value per particle column
            b = torch.as_tensor(bw_silverman(init_particles.flatten(1, -1), bw_scale), dtype=torch.float).reshape(-1)
            if b.numel() != 1:
                bw_vec = b.numpy().copy()  # `bw ** 2 * torch.eye(P)` (mpf.py:36) -> covariance diag(bw_p^2)
            bw = float(b[0])
        model = likelihood.model
        kw = dict(model=model.family, uncertain_params=tuple(model.uncertain_params), log_space=bool(likelihood.log_space),
                  obs_std=float(likelihood.sigma), lr=float(opt_args.get("lr", 1e-3)), bw_scale=float(bw_scale), init_bw=float(bw), dt=model.dt)
        for k in ("g", "mass", "length"):
            if k in model.params_dict:
                kw[k] = float(model.params_dict[k])
        grid = None
        if model.family == "particle":
            kw.update(max_speed=float(model._max_speed), max_accel=float(model._max_acc), can_crash=bool(model.can_crash),
                      with_obstacle=bool(model.with_obstacle), cell_size=float(model.map_cell_size or 0.1),
                      # the one-step prediction runs Particle.step in full (likelihoods.py:30-46 -> particle.py:145-153)
                      control_type=str(model.control_type), deterministic=bool(model.deterministic),
                      noise_std=tuple(float(v) for v in torch.as_tensor(model.dyn_std, dtype=torch.float).reshape(-1).expand(2)))
            grid = model.obst_map.map.astype(np.float32) if model.obst_map is not None else None
        kw.update(opt_kw)
        self._dev = MpfContext(init_particles.numpy(), likelihood.loc.numpy(), grid=grid, **kw)
        if bw_vec is not None:
            self._dev.set_prior_bw(bw_vec)
        self.prior = _DevicePrior(self)
        # recorded control-noise draws for reproducible runs: an object whose next_mpf_noise() returns the next [da] draw or None
        # (see MultiDISCO.draw_source); None: the library's own generator
        self.draw_source = None

    def __deepcopy__(self, memo):
        import copy

        new = copy.copy(self)
        memo[id(self)] = new
        new.likelihood = copy.deepcopy(self.likelihood, memo)
        new._dev = self._dev.clone()
        new.prior = _DevicePrior(new)
        return new

    @property
    def x(self):
        return torch.from_numpy(self._dev.get_particles())

    def _feed_ctrl_noise(self, n):
        if self.draw_source is None:
            return
        rec = [self.draw_source.next_mpf_noise() for _ in range(n)]
        if rec and rec[0] is not None:
            self._dev.set_ctrl_noise(np.stack([np.asarray(r, np.float32).reshape(-1) for r in rec]))

    def phi(self, bw):
        self._feed_ctrl_noise(1)
        return torch.from_numpy(self._dev.phi(float(bw)))

    def optimize(self, action, new_obs, bw=None, n_steps=100, debug=False):  # mpf.py:64-86
        if new_obs is not None:
            self.likelihood.condition(action, new_obs)
        if bw is None:
            bw = silvermans_rule(self.x.view(-1, 1).numpy()) * self.bw_scale
        a = None if action is None else torch.as_tensor(action, dtype=torch.float).reshape(-1).numpy()
        o = None if new_obs is None else torch.as_tensor(new_obs, dtype=torch.float).reshape(-1).numpy()
        self._feed_ctrl_noise(int(n_steps))
        grads = self._dev.optimize(a, o, float(bw), int(n_steps))
        return torch.as_tensor(grads), bw
