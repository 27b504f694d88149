"""Import shim for the *reference* package (lubaroli/dust), used ONLY in the build container.

TEST INFRASTRUCTURE - not part of the product path.  `/root/reference` does not exist on the GPU box; this
module is imported only by `tests/golden/make_golden.py` (fixture generation) and by tests that are skipped
when the reference tree is absent.

The reference needs three things this image lacks:
  1. `numpy.float` (removed in numpy>=1.24; used as a default arg at dust/utils/helper.py:90),
  2. `gpytorch.kernels.RBFKernel` (dust/inference/svmpc.py:2; gpytorch 1.5.0 pinned, environment.yaml:35),
  3. `KDEpy.bw_selection.silvermans_rule` (dust/inference/svmpc.py:3, mpf.py:3; KDEpy 1.1.0, environment.yaml:128).

(2) and (3) are third-party packages that are NOT vendored in /root/reference, so their arithmetic is restated
here from their published algorithms.  PARITY UNPINNED at these two boundaries: no reference test or golden
vector pins them (the reference has no tests at all).

gpytorch 1.5.0 RBFKernel.forward (x1 requires grad => the non-fused branch):
    x1_ = x1 / lengthscale ; x2_ = x2 / lengthscale
    sq_dist: subtract x1_.mean(-2) from both, then [-2 x1, |x1|^2, 1] @ [x2, 1, |x2|^2]^T, clamp_min(0)
    K = exp(-sq/2)
  lengthscale = softplus(raw_lengthscale = 0) = ln 2.  svmpc.py:78 assigns the misspelt attribute
  `lenghtscale`, so the bandwidth argument never reaches the kernel (plain attribute on our stand-in too).

KDEpy 1.1.0 silvermans_rule(data[obs,1]):  sigma = min(std(ddof=1), IQR/1.3489795) (the positive one when one of
them is 0), bw = sigma * (n*3/4)^(-1/5).
"""
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"
LN2 = float(torch.nn.functional.softplus(torch.zeros(())))


class _Evaluated:
    def __init__(self, t):
        self._t = t

    def evaluate(self):
        return self._t


class RBFKernel:
    """Stand-in with gpytorch-1.5.0 RBFKernel semantics (see module docstring)."""

    def __init__(self):
        self.raw_lengthscale = torch.zeros(1, 1)

    @property
    def lengthscale(self):
        return torch.nn.functional.softplus(self.raw_lengthscale)

    def forward(self, x1, x2):
        ls = self.lengthscale
        x1_ = x1.div(ls)
        x2_ = x2.div(ls)
        adjustment = x1_.mean(-2, keepdim=True)
        x1_ = x1_ - adjustment
        x2_ = x2_ - adjustment
        x1_norm = x1_.pow(2).sum(dim=-1, keepdim=True)
        x1_pad = torch.ones_like(x1_norm)
        x2_norm = x2_.pow(2).sum(dim=-1, keepdim=True)
        x2_pad = torch.ones_like(x2_norm)
        a = torch.cat([-2.0 * x1_, x1_norm, x1_pad], dim=-1)
        b = torch.cat([x2_, x2_pad, x2_norm], dim=-1)
        res = a.matmul(b.transpose(-2, -1))
        res = res.clamp_min(0)
        return res.div(-2).exp()

    def __call__(self, x1, x2=None):
        if x2 is None:
            x2 = x1
        return _Evaluated(self.forward(x1, x2))


def silvermans_rule(data):
    data = np.asarray(data, dtype=np.float64)
    assert data.ndim == 2 and data.shape[1] == 1
    obs = data.shape[0]
    if obs == 1:
        return 1.0
    if obs < 1:
        raise ValueError("Data must be of length > 0.")
    std = np.std(data, ddof=1)
    q75, q25 = np.percentile(data, [75, 25])
    iqr = (q75 - q25) / 1.3489795003921634
    sigma = min(std, iqr)
    if not sigma > 0:
        sigma = max(std, iqr)
    if sigma > 0:
        return float(sigma * (obs * 3 / 4.0) ** (-1 / 5))
    return 1.0


_installed = False


def install():
    """Make `import dust` resolve to /root/reference with the three shims in place."""
    global _installed
    if _installed:
        return
    import os

    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present (only available in the build container)")
    if not hasattr(np, "float"):
        np.float = float
    gp = types.ModuleType("gpytorch")
    gk = types.ModuleType("gpytorch.kernels")
    gk.RBFKernel = RBFKernel
    gp.kernels = gk
    sys.modules["gpytorch"] = gp
    sys.modules["gpytorch.kernels"] = gk
    kd = types.ModuleType("KDEpy")
    kb = types.ModuleType("KDEpy.bw_selection")
    kb.silvermans_rule = silvermans_rule
    kd.bw_selection = kb
    sys.modules["KDEpy"] = kd
    sys.modules["KDEpy.bw_selection"] = kb
    import matplotlib

    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True
