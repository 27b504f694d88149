"""Merwe scaled unscented transform with the reference's constructor and attributes (dust/utils/utf.py:7-145).

Host code by nature (a (2n+1)-point rule over an n <= 4 dimensional parameter distribution): MultiDISCO computes the sigma
points here and hands them, with `loc_weights`, to the device, where they are the dynamics samples of the rollouts."""
import torch


class MerweScaledUTF:
    def __init__(self, n, alpha=1e-3, beta=2.0, kappa=0.0, sqrt_method=None):
        self.n, self.pts = n, 2 * n + 1
        self.alpha, self.beta, self.kappa = alpha, beta, kappa
        # upper-triangular square root U with U^T U = A, as the reference's default (utf.py:51-57)
        self.sqrt = sqrt_method or (lambda A: torch.linalg.cholesky(A.transpose(-2, -1).conj()).transpose(-2, -1).conj())
        lam = alpha ** 2 * (n + kappa) - n
        c = 0.5 / (n + lam)
        self._loc = torch.full((self.pts,), c, dtype=torch.float)
        self._cov = torch.full((self.pts,), c, dtype=torch.float)
        self._cov[0] = lam / (n + lam) + (1 - alpha ** 2 + beta)
        self._loc[0] = lam / (n + lam)

    loc_weights = property(lambda self: self._loc)
    cov_weights = property(lambda self: self._cov)

    def compute_sigma_points(self, mu, K):
        """[n, 2n+1]: the mean, then mean +- the columns of sqrt((lambda + n) K) (utf.py:93-123)."""
        mu = torch.as_tensor(mu, dtype=torch.float)
        K = torch.as_tensor(K, dtype=torch.float)
        if self.n != mu.size(0):
            raise ValueError("expected size(x) {}, but size is {}".format(self.n, mu.size(0)))
        lam = self.alpha ** 2 * (self.n + self.kappa) - self.n
        U = self.sqrt((lam + self.n) * K)
        sig = torch.zeros(self.n, self.pts, dtype=torch.float)
        sig[:, 0] = mu
        sig[:, 1:self.n + 1] = U + mu.view(-1, 1)
        sig[:, self.n + 1:] = -U + mu.view(-1, 1)
        return sig

    def unscented_transform(self, sigmas):
        mu = sigmas @ self._loc
        y = sigmas - mu.view(-1, 1)
        return mu, y @ torch.diag(self._cov) @ y.t()
