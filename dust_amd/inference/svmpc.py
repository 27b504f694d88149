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
        if roll_strategy not in ("repeat", "mean", "resample"):
            raise ValueError("{} is an invalid roll strategy.".format(roll_strategy))
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
        if not torch.equal(cov, torch.diag(torch.diag(cov))) and cov.shape[-1] != 2:
            raise NotImplementedError("a full prior covariance has a HIP kernel for dim_a = 2")
        ctrl = likelihood.controller
        ctrl._svmpc_cfg.update(opt, likelihood=likelihood.kind, alpha=float(likelihood.alpha), weighted_prior=bool(weighted_prior),
                               roll_strategy=roll_strategy, sigma_p=cov.diag().sqrt().numpy(), p_cov=cov.detach().numpy().copy(),
                               **kernel_config(kernel))
        self._theta0 = torch.as_tensor(init_particles, dtype=torch.float).detach().clone()
        self._prior0 = (comp.loc.detach().clone(), prior.mixture_distribution.probs.detach().clone())
        self._uploaded = False
        self._prior_obj, self._prior_cov = prior, cov
        self._prior_stale = False
        # attribute compatibility (simulations.py never touches it): torch's optimiser object over a placeholder - the optimiser
        # STATE lives on the device and restarts at every roll, as torch's does when roll() swaps the parameter tensor
        self.optimizer = optimizer_class(params=[torch.zeros(1, requires_grad=True)], **opt_args)

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

    @property
    def prior(self):
        """The GMM in force (svmpc.py:160-170): after a forward() / update_prior() its means are the current particles."""
        if self._prior_stale and self._uploaded:
            from .svgd import get_gmm

            means, probs = self._ctx().get_prior()
            self._prior_obj = get_gmm(torch.from_numpy(means), torch.from_numpy(probs), self._prior_cov)
            self._prior_stale = False
        return self._prior_obj

    @prior.setter
    def prior(self, p):
        self._prior_obj = p
        comp = p.component_distribution.base_dist
        self._prior_stale = False
        if self._uploaded:
            self._ctx().set_prior(comp.loc.detach().numpy(), p.mixture_distribution.probs.detach().numpy())
        else:
            self._prior0 = (comp.loc.detach().clone(), p.mixture_distribution.probs.detach().clone())

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
        ctrl = self.likelihood.controller
        n_steps = self.n_steps if n_steps is None else n_steps
        if eps is None and ctrl.draw_source is not None:  # recorded draws (MultiDISCO.draw_source), e.g. of a reference run
            rec = [ctrl._recorded("eps") for _ in range(n_steps)]
            if rec and rec[0] is not None:
                eps = np.stack([np.asarray(r, np.float32) for r in rec])
        params, lp = ctrl._sample_params(params_dist, n_steps)
        self.likelihood.params_log_p = lp
        ctrl._feed_ctrl_noise(ctx, n_steps)
        ctx.svmpc_optimize(self._state(state), n_steps, eps, params)

    # -- svmpc.py:128-200
    def _resample(self, ctx):
        if self.roll_strategy != "resample":
            return None
        # svmpc.py:148-150: the last action of a fresh prior sample per particle (torch's RNG, as in the reference)
        return self.prior.sample([self.n_particles])[..., -1, :].reshape(self.n_particles, -1).numpy()

    def get_weights(self, state, params_dist, fast_pred=True):
        ctx = self._ctx(params_dist)
        if not fast_pred:
            params, _ = self.likelihood.controller._sample_params(params_dist)
            ctx.likelihood_sample(self._state(state), None, None if params is None else params[0])
        return torch.from_numpy(ctx.svmpc_get_weights())

    def roll(self, steps=-1, strategy="repeat"):
        ctx = self._ctx()
        last = None
        if strategy == "resample":
            last = self.prior.sample([self.n_particles])[..., -1, :].reshape(self.n_particles, -1).numpy()
        ctx.svmpc_roll(steps, strategy, last)

    def update_prior(self, weights=None):
        self._ctx().svmpc_update_prior(None if weights is None else torch.as_tensor(weights, dtype=torch.float).numpy())
        self._prior_stale = True

    def forward(self, state, params_dist, steps=-1, fast_pred=True):
        ctx = self._ctx(params_dist)
        if not fast_pred:
            params, _ = self.likelihood.controller._sample_params(params_dist)
            ctx.likelihood_sample(self._state(state), None, None if params is None else params[0])
        a_seq, pw = ctx.svmpc_forward_ex(steps, self._resample(ctx))
        self._prior_stale = True
        return torch.from_numpy(a_seq), torch.from_numpy(pw)

    def tick(self, state, params_dist, n_steps=None, eps=None):
        """optimize + forward enqueued back to back (dust_svmpc_tick)."""
        ctx = self._ctx(params_dist)
        n_steps = self.n_steps if n_steps is None else n_steps
        params, _ = self.likelihood.controller._sample_params(params_dist, n_steps)
        if self.roll_strategy == "resample":  # the last row is drawn on the host between the two halves
            ctx.svmpc_optimize(self._state(state), n_steps, eps, params)
            a_seq, pw = ctx.svmpc_forward_ex(-1, self._resample(ctx))
        else:
            a_seq, pw = ctx.svmpc_tick(self._state(state), n_steps, eps, params)
        self._prior_stale = True
        return torch.from_numpy(a_seq), torch.from_numpy(pw)
