"""Likelihood objects with the reference's signatures (dust/inference/likelihoods.py:12-135)."""
import numpy as np
import torch


class CostLikelihood:
    kind = None

    def __init__(self, n_samples, controller, model):
        self.n_samples = n_samples
        self.controller, self.model = controller, model
        self.last_costs = self.last_actions = self.last_states = self.last_policies = None
        self.params = self.params_log_p = None

    def sample(self, theta, state, params_dist, eps=None):
        """likelihoods.py:81-101: actions = theta + L eps, then MultiDISCO.forward.  `eps` (optional, [S,N,H,da]) replays
        given standard-normal draws; by default they come from the device Philox stream."""
        c = self.controller
        c._svmpc_cfg.setdefault("likelihood", self.kind)
        c._svmpc_cfg["alpha"] = float(self.alpha)
        ctx = c._ensure_ctx(self.model, params_dist)
        params, self.params_log_p = c._sample_params(params_dist)
        st = torch.as_tensor(state, dtype=torch.float).reshape(-1).numpy()
        c._feed_ctrl_noise(ctx)
        # theta is an ARGUMENT here (the reference never writes SVMPC.theta or the optimiser state from sample())
        costs, actions = ctx.likelihood_sample_at(st, torch.as_tensor(theta, dtype=torch.float).detach().numpy(),
                                                  None if eps is None else np.asarray(eps, np.float32),
                                                  None if params is None else params[0], want_actions=True)
        self.last_costs, self.last_actions = torch.from_numpy(costs), torch.from_numpy(actions)
        return self.last_costs, self.last_actions


class ExpectedCost(CostLikelihood):
    kind = "ExpectedCost"

    def __init__(self, alpha, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha

    def log_prob(self, costs=None):  # likelihoods.py:113-119
        costs = self.last_costs if costs is None else costs
        return -self.alpha * costs.mean(dim=0)


class ExponentiatedUtility(CostLikelihood):
    kind = "ExponentiatedUtility"

    def __init__(self, alpha, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha

    def log_prob(self, costs=None):  # likelihoods.py:127-135
        costs = self.last_costs if costs is None else costs
        return (-self.alpha * costs).logsumexp(0) - torch.as_tensor(costs.size(0), dtype=torch.float).log()


class GaussianLikelihood:
    """One-step prediction likelihood of the dynamics filter (likelihoods.py:12-64); MPF evaluates it on the device."""

    def __init__(self, initial_obs, obs_std, model, log_space=False):
        initial_obs = torch.as_tensor(initial_obs, dtype=torch.float)
        assert initial_obs.ndim == 1, "Gaussian likelihood needs a single dimensional loc tensor."
        self.dim = initial_obs.shape[0]
        self.sigma = obs_std
        self.model, self.log_space = model, log_space
        self.loc, self.past_obs, self.past_action = initial_obs, None, None

    def condition(self, action, new_obs, covariance_matrix=None):
        self.past_obs, self.loc, self.past_action = self.loc, torch.as_tensor(new_obs, dtype=torch.float).reshape(-1), action

    @property
    def density(self):  # likelihoods.py:62-64
        import torch.distributions as dist

        return dist.MultivariateNormal(self.loc, self.sigma ** 2 * torch.eye(self.dim))

    def sample(self, theta):
        """likelihoods.py:30-46: one-step predictions of the model under each parameter particle (host side: the MPF kernel
        evaluates the same step and its parameter Jacobian on the device; this is the object-protocol form)."""
        assert self.past_action is not None, "Previous action is None. Need at least one observation to start sampling."
        theta = torch.as_tensor(theta, dtype=torch.float)
        params = theta.exp() if self.log_space else theta
        params_dict = self.model.params_to_dict(params)
        states = self.past_obs.repeat(theta.shape[0], 1)
        return self.model.step(states, torch.as_tensor(self.past_action, dtype=torch.float), params_dict)

    def log_prob(self, samples):  # likelihoods.py:48-49
        return self.density.log_prob(torch.as_tensor(samples, dtype=torch.float)).unsqueeze(-1)
