"""`get_gmm`, `bw_silverman` (dust/inference/svgd.py:10-89) - host-side helpers with the reference's signatures."""
import numpy as np
import torch
import torch.distributions as dist


def get_gmm(x, weights, covariance):
    """svgd.py:84-89: MixtureSameFamily(Categorical(weights), Independent(MVN(x, covariance), 1)).  SVMPC reads the
    means / mixture weights / covariance diagonal out of it and hands them to the device."""
    mix = dist.Categorical(weights)
    comp = dist.Independent(dist.MultivariateNormal(x.detach(), covariance), 1)
    return dist.mixture_same_family.MixtureSameFamily(mix, comp)


def bw_silverman(x, bw_scale=1.0):
    """svgd.py:10-81: 0.9 * A * n^(-1/5), A = IQR/1.349 if 0 < IQR < min(std) else std (per column)."""
    x = torch.as_tensor(x, dtype=torch.float)
    flat = x.numpy().reshape(-1)
    iqr = (np.percentile(flat, 75) - np.percentile(flat, 25)) / 1.349
    std = torch.std(x, axis=0)
    a = iqr if (iqr > 0 and iqr < std.min()) else std
    return bw_scale * (0.9 * a * len(x) ** (-0.2))
