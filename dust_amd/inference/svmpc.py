"""`SVMPC` with the reference's constructor and methods (dust/inference/svmpc.py:14-200, svgd.py:109-125)."""
import numpy as np
import torch

from ..kernels import kernel_config


class SVMPC:
    def __init__(self, init_particles, prior, likelihood, roll_strategy="repeat", weighted_prior=False, kernel=None, bw_scale=1.0,
                 n_particles=None, n_steps=100, optimizer_class=torch.optim.Adam, **opt_args):
        self.likelihood = likelihood
        self.kernel, self.bw_scale = kernel, bw_scale
        self.n_particles = n_particles if n_particles is not None else init_particles.shape[0]
        self.n_steps = n_steps
        self.optimizer_class, self.opt_args = optimizer_class, opt_args
        self.w_prior, self.roll_strategy = weighted_prior, roll_strategy
        if roll_strategy not in ("repeat", "mean"):
            raise NotImplementedError("roll strategy %r has no HIP kernel ('resample' draws from the prior on the host)" % roll_strategy)
        if optimizer_class is torch.optim.SGD:
            opt = dict(optimizer="SGD", lr=float(opt_args.get("lr", 1e-3)))
        elif optimizer_class is torch.optim.Adam:
            b = opt_args.get("betas", (0.9, 0.999))
            opt = dict(optimizer="Adam", lr=float(opt_args.get("lr", 1e-3)), adam=(b[0], b[1], opt_args.get("eps", 1e-8)))
        else:
            raise NotImplementedError("optimizer %r has no HIP kernel (SGD and Adam do)" % (optimizer_class,))
        comp = prior.component_distribution.base_dist
        cov = comp.covariance_matrix
        cov = cov.reshape(-1, cov.shape[-2], cov.shape[-1])[0]
        if not torch.equal(cov, torch.diag(torch.diag(cov))):
            raise NotImplementedError("only diagonal prior covariances have a HIP kernel")
        ctrl = likelihood.controller
        ctrl._svmpc_cfg.update(opt, likelihood=likelihood.kind, alpha=float(likelihood.alpha), weighted_prior=bool(weighted_prior),
                               roll_strategy=roll_strategy, sigma_p=cov.diag().sqrt().numpy(), **kernel_config(kernel))
        self._theta0 = torch.as_tensor(init_particles, dtype=torch.float).detach().clone()
        self._prior0 = (comp.loc.detach().clone(), prior.mixture_distribution.probs.detach().clone())
        self._uploaded = False
        self.prior = prior

    # -- device plumbing
    def _ctx(self, params_dist=None):
        ctrl = self.likelihood.controller
        ctx = ctrl._ensure_ctx(self.likelihood.model, params_dist)
        if not self._uploaded:
            ctx.set_theta(self._theta0.numpy())
            ctx.set_prior(self._prior0[0].numpy(), self._prior0[1].numpy())
            self._uploaded = True
        return ctx

    def __deepcopy__(self, memo):
        import copy

        new = copy.copy(self)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, v if k in ("kernel", "optimizer_class") else copy.deepcopy(v, memo))
        if self._uploaded:  # the cloned controller context already carries theta / prior
            new._uploaded = True
        return new

    @property
    def theta(self):
        return torch.from_numpy(self._ctx().get_theta()) if self._uploaded else self._theta0

    @theta.setter
    def theta(self, v):
        self._theta0 = torch.as_tensor(v, dtype=torch.float).detach().clone()
        if self._uploaded:
            self._ctx().set_theta(self._theta0.numpy())

    # -- svmpc.py:32-85
    def phi(self, log_p, bw=None, sigma=None):
        ctx = self._ctx()
        _, costs, actions = log_p(self.theta)
        phi, _, _ = ctx.svmpc_phi(torch.as_tensor(costs).numpy(), torch.as_tensor(actions).numpy())
        return torch.from_numpy(phi)

    def _state(self, state):
        return torch.as_tensor(state, dtype=torch.float).reshape(-1).numpy()

    # -- svmpc.py:87-126
    def step(self, state, params_dist, bw=None, sigma=None, eps=None):
        self.optimize(state, params_dist, n_steps=1, eps=None if eps is None else np.asarray(eps, np.float32)[None])

    def optimize(self, state, params_dist, bw=None, n_steps=None, debug=False, eps=None):
        """`bw` is accepted and ignored, as in the reference (dead value on both kernel branches)."""
        ctx = self._ctx(params_dist)
        n_steps = self.n_steps if n_steps is None else n_steps
        params, lp = self.likelihood.controller._sample_params(params_dist, n_steps)
        self.likelihood.params_log_p = lp
        ctx.svmpc_optimize(self._state(state), n_steps, eps, params)

    # -- svmpc.py:128-200
    def forward(self, state, params_dist, steps=-1, fast_pred=True):
        if steps != -1:
            raise NotImplementedError("roll by %d steps: only steps=-1 (one control tick) is implemented" % steps)
        ctx = self._ctx(params_dist)
        if not fast_pred:
            params, _ = self.likelihood.controller._sample_params(params_dist)
            ctx.likelihood_sample(self._state(state), None, None if params is None else params[0])
        a_seq, pw = ctx.svmpc_forward()
        return torch.from_numpy(a_seq), torch.from_numpy(pw)

    def tick(self, state, params_dist, n_steps=None, eps=None):
        """optimize + forward enqueued back to back (dust_svmpc_tick)."""
        ctx = self._ctx(params_dist)
        n_steps = self.n_steps if n_steps is None else n_steps
        params, _ = self.likelihood.controller._sample_params(params_dist, n_steps)
        a_seq, pw = ctx.svmpc_tick(self._state(state), n_steps, eps, params)
        return torch.from_numpy(a_seq), torch.from_numpy(pw)
