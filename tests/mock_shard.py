"""CPU stand-in for dust_amd.parallel.DeviceShard (TEST INFRASTRUCTURE): same four phases and gather buffers, the
arithmetic done by the CPU oracle.  Lets the N>1 orchestration (phase order, in-place all-gathers, shard bounds) run under
gloo with world_size 2 in a container that has no GPU."""
import numpy as np
import torch

import oracle
from dust_amd.parallel import shard_bounds
from oracle import Oracle


class MockShard:
    def __init__(self, cfg, rank, world):
        self.cfg, self.rank, self.world = cfg, rank, world
        self.N, self.S, self.H = cfg["N"], cfg["S"], cfg["H"]
        self.off, self.n_loc = shard_bounds(self.N, rank, world)
        self.model = cfg.get("model", "pendulum")
        self.da = 1 if self.model == "pendulum" else 2
        self.M = cfg.get("M", 1)
        if self.model == "particle":  # the cfg4 family: 2-D point mass on the demo's occupancy grid, sampled mass
            self.o = Oracle(model="particle", N=self.N, S=self.S, M=self.M, H=self.H, uncertain_params=("mass",), mass=2.0,
                            grid=oracle.grid_4x4_map())
        else:
            self.o = Oracle(model="pendulum", N=self.N, S=self.S, M=1, H=self.H)
        self.D = self.H * self.da
        self.shard_elems = self.n_loc * self.D
        self.theta_all = torch.zeros(self.N * self.D)
        self.score_all = torch.zeros(self.N * self.D)
        self.lw_all = torch.zeros(self.N)
        self.sig = np.full(self.da, cfg["sigma_a"], np.float32)
        self.sp = np.full(self.da, cfg["sigma_p"], np.float32)
        self.aliased = False
        self.a_seq = None
        self.pw = None

    def _rows(self, full):
        return full[self.off * self.D:(self.off + self.n_loc) * self.D]

    def set_state(self, theta, mu, a_mat=None, mix=None):
        self.theta_all[:] = torch.from_numpy(np.asarray(theta, np.float32).reshape(-1))
        self.mu = np.asarray(mu, np.float32).reshape(self.N, self.H, self.da).copy()
        self.mix = np.ones(self.N, np.float32)

    def _theta(self):
        return self.theta_all.numpy().reshape(self.N, self.H, self.da)

    def local_score(self, state, eps=None, params=None):
        th = self._theta()
        mu = th if self.aliased else self.mu
        actions = self.o.sample_actions(th, eps, self.sig)
        self.costs = self.o.rollout_cost(state, actions, params)
        _, _, sc = self.o.score(th, mu, self.mix, self.sp, self.costs, actions, self.cfg["alpha"], self.sig)
        self._rows(self.score_all)[:] = torch.from_numpy(sc.reshape(-1))[self.off * self.D:(self.off + self.n_loc) * self.D]

    def local_rollout(self, state, eps=None, params=None):
        th = self._theta()  # the other shards' rows may still be in flight here: only the local rows are used below
        self._actions = self.o.sample_actions(th, eps, self.sig)
        self.costs = self.o.rollout_cost(state, self._actions, params)
        self._local_rows = th[self.off:self.off + self.n_loc].copy()

    def local_prior_score(self):
        th = self._theta()
        lo, hi = self.off, self.off + self.n_loc
        assert np.array_equal(th[lo:hi], self._local_rows)  # the all-gather never touches a rank's own rows
        mu = th if self.aliased else self.mu
        _, _, sc = self.o.score(th, mu, self.mix, self.sp, self.costs, self._actions, self.cfg["alpha"], self.sig)
        self._rows(self.score_all)[:] = torch.from_numpy(sc.reshape(-1))[lo * self.D:hi * self.D]

    def apply_phi(self):
        th = self._theta()
        phi = self.o.phi_k1(th, self.score_all.numpy().reshape(self.N, self.H, self.da))
        new = Oracle.sgd(th, phi, self.cfg["lr"]).reshape(-1)
        self._rows(self.theta_all)[:] = torch.from_numpy(new)[self.off * self.D:(self.off + self.n_loc) * self.D]

    def forward_local(self):
        th = self._theta()
        mu = th if self.aliased else self.mu
        r = self.o.forward(self.costs, th, mu, self.mix, self.sp, self.cfg["alpha"])
        lw = r["log_l"] + r["log_p"]
        self.lw_all[self.off:self.off + self.n_loc] = torch.from_numpy(lw[self.off:self.off + self.n_loc])

    def forward_finish(self, want_outputs=False):
        lw = self.lw_all.numpy().astype(np.float64)
        p = np.exp(lw - lw.max())
        p /= p.sum()
        th = self._theta().copy()
        best = int(np.argmax(p))
        a_seq = th[best].copy()
        rolled = np.concatenate([th[:, 1:], th[:, -1:]], axis=1).reshape(-1)
        self._rows(self.theta_all)[:] = torch.from_numpy(rolled)[self.off * self.D:(self.off + self.n_loc) * self.D]
        self.aliased = True
        return (a_seq, p.astype(np.float32)) if want_outputs else (None, None)

    def sync(self):
        pass
