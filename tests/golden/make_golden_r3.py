#!/usr/bin/env python3
"""Round-3 golden: MPF(bw=None) with P = 2 parameters - the reference's own `MPF.__init__` / `update_prior(None)` (mpf.py:29-38):
`bw_silverman(x.flatten(1, -1))` (svgd.py:55-81, `_select_sigma` svgd.py:10-25) returns one bandwidth PER COLUMN when the pooled
IQR / 1.349 is not below every column's standard deviation, and `bw ** 2 * torch.eye(P)` turns the vector into the covariance
diag(bw_p^2) of the FIRST prior.  Recorded: the bandwidth vector, the prior's covariance diagonal and log-density at probe points,
`phi(bw)` on the conditioned likelihood, and two `optimize(..., bw=0.08)` calls (after the first, update_prior(bw) makes the prior
isotropic again - mpf.py:85).  An explicit bw goes to optimize(): the `bw=None` branch THERE calls KDEpy (absent).
TEST INFRASTRUCTURE - needs /root/reference:   python tests/golden/make_golden_r3.py

Two particle sets: `mpf_bwvec` (columns of different location and spread: pooled IQR / 1.349 >= min column std -> the per-column
std branch) and `mpf_bwiqr` (tight columns of one location with outliers: the pooled-IQR branch -> a scalar)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the shim, imports torch and the reference)

import torch  # noqa: E402
import torch.distributions as dist  # noqa: E402
from dust.inference.likelihoods import GaussianLikelihood  # noqa: E402
from dust.inference.mpf import MPF  # noqa: E402
from dust.inference.svgd import bw_silverman  # noqa: E402
from dust.models.pendulum import PendulumModel  # noqa: E402


def run(tag, x0, n_steps=4, bw_opt=0.08, lr=0.001):
    model = PendulumModel(uncertain_params=("length", "mass"))
    obs0 = torch.tensor([3.0, 0.0])
    action = torch.tensor(1.3)
    true_model = PendulumModel(length=0.9, mass=1.1)
    obs1 = true_model.step(obs0.view(1, -1), action.view(1, 1)).view(-1)
    lik = GaussianLikelihood(initial_obs=obs0, obs_std=0.1, model=model, log_space=False)
    mpf = MPF(init_particles=x0.clone(), likelihood=lik, optimizer_class=torch.optim.SGD, lr=lr, bw=None, bw_scale=1.0)
    bw0 = torch.as_tensor(bw_silverman(x0.flatten(1, -1), 1.0), dtype=torch.float).reshape(-1)
    cov = mpf.prior.component_distribution.base_dist.covariance_matrix[0]
    probe = torch.stack([torch.linspace(0.4, 1.6, 9), torch.linspace(1.5, 0.5, 9)], dim=1).contiguous()
    g = dict(Mp=x0.shape[0], P=2, n_steps=n_steps, bw_opt=bw_opt, lr=lr, obs_std=0.1, x0=mg.npf(x0), obs0=mg.npf(obs0), obs1=mg.npf(obs1),
             action=mg.npf(action).reshape(-1), bw_init=mg.npf(bw0), prior_cov_diag=mg.npf(torch.diagonal(cov)),
             prior_cov_offdiag_max=np.float32(float((cov - torch.diag(torch.diagonal(cov))).abs().max())),
             probe=mg.npf(probe), probe_log_prob0=mg.npf(mpf.prior.log_prob(probe)))
    lik.condition(action, obs1)
    g["phi0"] = mg.npf(mpf.phi(bw_opt))
    lik.loc = obs0
    lik.past_obs = None
    lik.past_action = None
    grads, _ = mpf.optimize(action, obs1, bw=bw_opt, n_steps=n_steps)
    g.update(x_final=mg.npf(mpf.x), grad_norms=mg.npf(grads), probe_log_prob1=mg.npf(mpf.prior.log_prob(probe)))
    action2 = action * 0.5
    obs2 = true_model.step(obs1.view(1, -1), action2.view(1, 1)).view(-1)
    grads2, _ = mpf.optimize(action2, obs2, bw=bw_opt, n_steps=n_steps)
    g.update(action2=mg.npf(action2).reshape(-1), obs2=mg.npf(obs2), x_final2=mg.npf(mpf.x), grad_norms2=mg.npf(grads2))
    np.savez_compressed(os.path.join(mg.OUT, tag + ".npz"), **g)
    print("wrote %s: bw_init %s" % (tag, bw0.numpy()))
    return g


if __name__ == "__main__":
    torch.manual_seed(11)
    # heavy-tailed columns of different scale: pooled IQR / 1.349 is NOT below the smaller column std -> per-column std
    x = torch.stack([0.9 + 0.02 * dist.StudentT(2.0).sample([16]).clamp(-6, 6), 1.0 + 0.25 * torch.randn(16)], dim=1).clamp(min=0.3)
    g = run("mpf_bwvec", x)
    assert g["bw_init"].size == 2 and abs(g["bw_init"][0] - g["bw_init"][1]) > 1e-3, g["bw_init"]
    # tight columns of one location with a few outliers: the pooled IQR / 1.349 is below both column stds -> a scalar
    x = 1.0 + 0.02 * torch.randn(16, 2)
    x[:2] += 0.4  # two outliers per column: the column stds are ~0.14, the pooled IQR / 1.349 ~0.02
    x[2:4] -= 0.35
    g = run("mpf_bwiqr", x)
    assert g["bw_init"].size == 1, g["bw_init"]
