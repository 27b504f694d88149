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
        ctx.set_theta(torch.as_tensor(theta, dtype=torch.float).detach().numpy())
        params, self.params_log_p = c._sample_params(params_dist)
        st = torch.as_tensor(state, dtype=torch.float).reshape(-1).numpy()
        costs, actions = ctx.likelihood_sample(st, None if eps is None else np.asarray(eps, np.float32),
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
