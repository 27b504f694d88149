"""Particle-sharded SVGD-MPC over the GPUs of one node (one process per GPU, RCCL over xGMI via torch.distributed).

The policy index n is data-parallel: rank g owns N/G Stein particles with all their S*M rollouts,
costs, softmax weights, likelihood score, a_mat rows, optimiser update and roll - no communication.  The pairwise stages
need every particle, so per SVGD iteration the ranks all-gather, IN PLACE in the context-owned [N][D] buffers,
  (1) theta  (after the optimiser update; the prior means alias theta from the second tick on, so the prior pass needs it),
  (2) score  (after the local score, before the Gram / phi kernel - the exchange BASELINE.json's north_star names),
and once per tick the N log-weights.  xGMI is a point-to-point mesh and these messages are a few hundred KB at most, so
the collectives are latency-bound; they are issued on the context's stream so no host synchronisation is needed.

`backend` is anything with the small collective interface below (torch.distributed for RCCL/gloo); the device pointers
are wrapped as torch tensors through __cuda_array_interface__ without copying.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .backend import Context


class _DevBuf:
    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def shard_bounds(n_total, rank, world):
    """Contiguous equal shards of the particle index (N must divide evenly, as the in-place all-gather requires)."""
    if n_total % world:
        raise ValueError("n_particles (%d) must be divisible by the number of GPUs (%d)" % (n_total, world))
    n_loc = n_total // world
    return rank * n_loc, n_loc


class ShardedSVMPC:
    def __init__(self, common_cfg, rank, world, dist):
        import torch

        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        off, n_loc = shard_bounds(common_cfg["N"], rank, world)
        self.off, self.n_loc = off, n_loc
        self.ctx = Context(**dict(common_cfg, shard_offset=off, shard_size=n_loc))
        lib = L.load()
        L.check(lib.dust_set_stream(self.ctx._h, L.VP(torch.cuda.current_stream().cuda_stream)))
        th, sc, nb = L.VP(), L.VP(), C.c_size_t(0)
        L.check(lib.dust_gather_buffers(self.ctx._h, C.byref(th), C.byref(sc), C.byref(nb)))
        nd = self.ctx.N * self.ctx.D
        dev = torch.device("cuda", torch.cuda.current_device())
        self.theta_all = torch.as_tensor(_DevBuf(th.value, nd), device=dev)
        self.score_all = torch.as_tensor(_DevBuf(sc.value, nd), device=dev)
        lw, nb2 = L.VP(), C.c_size_t(0)
        self._lw_ptr = None
        self._shard = n_loc * self.ctx.D

    def _gather(self, full, shard_elems):
        lo = self.rank * shard_elems
        self.dist.all_gather_into_tensor(full, full[lo:lo + shard_elems])

    def set_state(self, theta, mu, a_mat=None):
        self.ctx.set_theta(theta)
        self.ctx.set_prior(mu)
        self.ctx.set_a_mat(theta if a_mat is None else a_mat)

    def tick(self, state, n_iters, eps=None):
        lib, h = L.load(), self.ctx._h
        st = np.ascontiguousarray(np.asarray(state, np.float32).reshape(-1))
        stp = st.ctypes.data_as(L.FP)
        for k in range(n_iters):
            e = None
            if eps is not None:
                ek = np.ascontiguousarray(eps[k], dtype=np.float32)
                e = C.cast(ek.ctypes.data_as(L.FP), L.VP)
            L.check(lib.dust_svmpc_local_score(h, stp, e, None, 0))
            self._gather(self.score_all, self._shard)
            L.check(lib.dust_svmpc_apply_phi(h))
            self._gather(self.theta_all, self._shard)
        lw, nb = L.VP(), C.c_size_t(0)
        L.check(lib.dust_svmpc_forward_local(h, C.byref(lw), C.byref(nb)))
        if self._lw_ptr != lw.value:
            self._lw_ptr = lw.value
            self.lw_all = self.torch.as_tensor(_DevBuf(lw.value, self.ctx.N), device=self.theta_all.device)
        self._gather(self.lw_all, self.n_loc)
        L.check(lib.dust_svmpc_forward_finish(h, None, None))
        self._gather(self.theta_all, self._shard)  # rolled rows of the other shards

    def outputs(self):
        a_seq = np.empty((self.ctx.H, self.ctx.da), np.float32)
        pw = np.empty(self.ctx.N, np.float32)
        return a_seq, pw

    def sync(self):
        self.torch.cuda.current_stream().synchronize()
