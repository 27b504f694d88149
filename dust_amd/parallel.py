"""Particle-sharded SVGD-MPC over the GPUs of one node (one process per GPU, RCCL over xGMI via torch.distributed).

The policy index n is data-parallel: rank g owns N/G Stein particles with all their S*M rollouts,
costs, softmax weights, likelihood score, a_mat rows, optimiser update and roll - no communication.  The pairwise stages
need every particle, so per SVGD iteration the ranks all-gather, IN PLACE in the context-owned [N][D] buffers,
  (1) score  (after the local score, before the Gram / phi kernel - the exchange BASELINE.json's north_star names),
  (2) theta  (after the optimiser update: from the second tick on the prior means alias theta, so the next prior pass and
              the next Gram pass both need every shard's new particles; tick(overlap=True) starts it asynchronously and
              runs the next iteration's rollouts - which read only the rank's own particles - underneath it),
and once per tick the N log-weights (then the rolled theta).  xGMI is a point-to-point mesh and these messages are a few
hundred KB at most, so the collectives are latency-bound; they are issued on the stream the kernels run on, so no host
synchronisation is needed between kernels and collectives.

Structure: a *shard* object (DeviceShard for the HIP library; tests inject a CPU stand-in) exposes the four local phases
and its three gather buffers as torch tensors; `tick()` is the fixed phase/collective order; `comm` performs the in-place
all-gather (TorchComm over torch.distributed - RCCL on GPUs, gloo in the CPU tests - or LocalComm, which runs several
shards inside one process by copying slices, used to check sharded == unsharded on a single GPU).
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
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %d of %d" % (rank, world))
    if n_total % world:
        raise ValueError("n_particles (%d) must be divisible by the number of GPUs (%d)" % (n_total, world))
    n_loc = n_total // world
    return rank * n_loc, n_loc


class DeviceShard:
    """One rank's slice of the problem on one GPU: a sharded dust_ctx plus torch views of its gather buffers."""

    def __init__(self, common_cfg, rank, world, device_index=None, use_torch_stream=True):
        import torch

        self.torch = torch
        off, n_loc = shard_bounds(common_cfg["N"], rank, world)
        self.rank, self.world, self.off, self.n_loc = rank, world, off, n_loc
        cfg = dict(common_cfg, shard_offset=off, shard_size=n_loc)
        if device_index is not None:
            cfg["device"] = device_index
        self.ctx = Context(**cfg)
        lib = L.load()
        dev = torch.device("cuda", cfg.get("device", 0))
        if use_torch_stream:
            with torch.cuda.device(dev):
                L.check(lib.dust_set_stream(self.ctx._h, L.VP(torch.cuda.current_stream().cuda_stream)))
        th, sc, nb = L.VP(), L.VP(), C.c_size_t(0)
        L.check(lib.dust_gather_buffers(self.ctx._h, C.byref(th), C.byref(sc), C.byref(nb)))
        nd = self.ctx.N * self.ctx.D
        self.theta_all = torch.as_tensor(_DevBuf(th.value, nd), device=dev)
        self.score_all = torch.as_tensor(_DevBuf(sc.value, nd), device=dev)
        self.lw_all = None
        self._lw_ptr = None
        self.shard_elems = n_loc * self.ctx.D
        self.N, self.D = self.ctx.N, self.ctx.D

    def set_state(self, theta, mu, a_mat=None, mix=None):
        self.ctx.set_theta(theta)
        self.ctx.set_prior(mu, mix)
        self.ctx.set_a_mat(theta if a_mat is None else a_mat)

    # -- the four local phases (each only enqueues kernels)
    def _args(self, state, eps, params):
        st = np.ascontiguousarray(np.asarray(state, np.float32).reshape(-1))
        e = None
        if eps is not None:
            ek = np.ascontiguousarray(eps, dtype=np.float32)
            e = C.cast(ek.ctypes.data_as(L.FP), L.VP)
        p = None
        if params is not None:
            pk = np.ascontiguousarray(params, dtype=np.float32)
            p = pk.ctypes.data_as(L.FP)
        return st, e, p, (ek if eps is not None else None, pk if params is not None else None)

    def local_score(self, state, eps=None, params=None):
        st, e, p, _keep = self._args(state, eps, params)
        L.check(L.load().dust_svmpc_local_score(self.ctx._h, st.ctypes.data_as(L.FP), e, p, 0))

    def local_rollout(self, state, eps=None, params=None):
        """Rollout / likelihood half of the local score: reads only this shard's particles."""
        st, e, p, _keep = self._args(state, eps, params)
        L.check(L.load().dust_svmpc_local_rollout(self.ctx._h, st.ctypes.data_as(L.FP), e, p, 0))

    def local_prior_score(self):
        """Prior half (reads every particle): score rows of this shard = grad_lik + grad_pri."""
        L.check(L.load().dust_svmpc_local_prior_score(self.ctx._h))

    def apply_phi(self):
        L.check(L.load().dust_svmpc_apply_phi(self.ctx._h))

    def forward_local(self):
        lw, nb = L.VP(), C.c_size_t(0)
        L.check(L.load().dust_svmpc_forward_local(self.ctx._h, C.byref(lw), C.byref(nb)))
        if self._lw_ptr != lw.value:
            self._lw_ptr = lw.value
            self.lw_all = self.torch.as_tensor(_DevBuf(lw.value, self.ctx.N), device=self.theta_all.device)

    def forward_finish(self, want_outputs=False):
        if not want_outputs:
            L.check(L.load().dust_svmpc_forward_finish(self.ctx._h, None, None))
            return None, None
        a_seq = np.empty((self.ctx.H, self.ctx.da), np.float32)
        pw = np.empty(self.ctx.N, np.float32)
        L.check(L.load().dust_svmpc_forward_finish(self.ctx._h, a_seq.ctypes.data_as(L.FP), pw.ctypes.data_as(L.FP)))
        return a_seq, pw

    def sync(self):
        self.ctx.sync()
        self.torch.cuda.synchronize(self.theta_all.device)


class TorchComm:
    """In-place all-gather over torch.distributed ("nccl" = RCCL on ROCm; "gloo" in the CPU tests)."""

    def __init__(self, dist, rank):
        self.dist, self.rank = dist, rank

    def all_gather_inplace(self, shards, name, shard_elems):
        (sh,) = shards
        full = getattr(sh, name)
        lo = self.rank * shard_elems
        if full.is_cuda:
            self.dist.all_gather_into_tensor(full, full[lo:lo + shard_elems])
        else:  # gloo has no all_gather_into_tensor for a view of the output: gather to a list of views
            world = self.dist.get_world_size()
            parts = [full[r * shard_elems:(r + 1) * shard_elems] for r in range(world)]
            self.dist.all_gather(parts, full[lo:lo + shard_elems].clone())


    def all_gather_start(self, shards, name, shard_elems):
        """Non-blocking form: returns a handle whose wait() makes the caller's stream (host, under gloo) wait for it."""
        (sh,) = shards
        full = getattr(sh, name)
        lo = self.rank * shard_elems
        if full.is_cuda:
            return self.dist.all_gather_into_tensor(full, full[lo:lo + shard_elems], async_op=True)
        self.all_gather_inplace(shards, name, shard_elems)
        return _Done()


class _Done:
    def wait(self):
        return True


class LocalComm:
    """Several shards inside ONE process: the all-gather is a set of slice copies (single-GPU equivalence tests)."""

    def all_gather_start(self, shards, name, shard_elems):
        self.all_gather_inplace(shards, name, shard_elems)
        return _Done()

    def all_gather_inplace(self, shards, name, shard_elems):
        for src in shards:
            lo = src.rank * shard_elems
            piece = getattr(src, name)[lo:lo + shard_elems]
            for dst in shards:
                if dst is not src:
                    getattr(dst, name)[lo:lo + shard_elems].copy_(piece)


def tick(shards, comm, state, n_iters, eps=None, params=None, want_outputs=False, overlap=False, final_gather=True):
    """One control tick = n_iters SVGD iterations + forward, for the shard(s) this process drives.

    `shards` is a 1-tuple under torch.distributed (one rank per process) or all shards under LocalComm.
    eps[k] / params[k] are the per-iteration noise [S][N][H][da] / dynamics samples (None: device Philox / no sampling).
    overlap=True runs the rollouts (which read only the rank's own particles) while the all-gather of theta is still in
    flight.  Measured at world size 1 on MI355X the extra host calls cost more than the overlap can give while this driver
    is host-stepped from Python (1 970 vs 2 820 ticks/s), so it is off by default; the split entry points are what a
    C-side driver with its own RCCL communicator would use.
    """
    if overlap:
        pending = None
        for k in range(n_iters):
            for sh in shards:
                sh.local_rollout(state, None if eps is None else eps[k], None if params is None else params[k])
            if pending is not None:
                pending.wait()  # theta of every shard has arrived
            for sh in shards:
                sh.local_prior_score()
            comm.all_gather_inplace(shards, "score_all", shards[0].shard_elems)
            for sh in shards:
                sh.apply_phi()
            pending = comm.all_gather_start(shards, "theta_all", shards[0].shard_elems)
        if pending is not None:
            pending.wait()
    else:
        for k in range(n_iters):
            for sh in shards:
                sh.local_score(state, None if eps is None else eps[k], None if params is None else params[k])
            comm.all_gather_inplace(shards, "score_all", shards[0].shard_elems)
            for sh in shards:
                sh.apply_phi()
            comm.all_gather_inplace(shards, "theta_all", shards[0].shard_elems)
    for sh in shards:
        sh.forward_local()
    comm.all_gather_inplace(shards, "lw_all", shards[0].n_loc)
    outs = [sh.forward_finish(want_outputs) for sh in shards]
    # rolled rows of the other shards.  (A GPU shard rolls ALL rows itself when the strategy is "repeat" / "mean" - it holds every
    # rank's particles at this point - which is what the C-side sharded tick relies on to end without this gather;
    # final_gather=False checks exactly that.)
    if final_gather:
        comm.all_gather_inplace(shards, "theta_all", shards[0].shard_elems)
    return outs[0]


class ShardedSVMPC:
    """One rank of a torch.distributed job.

    c_side=True (default on GPUs): the context owns an RCCL communicator (dust_comm_init) and libdust_amd runs the whole sharded
    tick itself - kernels and all-gathers on one stream, no Python between them; torch.distributed only carries the 128-byte
    communicator id from rank 0 to the others.  c_side=False: the phase / collective order is stepped from Python through
    torch.distributed (the form the gloo CPU tests and the single-GPU LocalComm equivalence tests drive)."""

    def __init__(self, common_cfg, rank, world, dist, c_side=True):
        self.c_side = bool(c_side)
        self.rank, self.world = rank, world
        if self.c_side:
            # ncclCommInitRank is a collective without a time-out: a rank that fails BEFORE it (bad shard, no device, no librccl) would
            # leave the others hanging inside it.  Every rank therefore runs the local checks first, the ranks exchange the outcome over
            # torch.distributed, and either all of them go on to the communicator or all of them raise (include/dust_amd.h, ABORT RULE).
            err = None
            try:
                off, n_loc = shard_bounds(common_cfg["N"], rank, world)
                self.ctx = Context(**dict(common_cfg, shard_offset=off, shard_size=n_loc))
                self.ctx.comm_validate(rank, world)
            except Exception as e:  # noqa: BLE001 - whatever went wrong, the other ranks must hear of it
                err = "rank %d: %s" % (rank, e)
            verdicts = [None] * world
            dist.all_gather_object(verdicts, err)
            bad = [v for v in verdicts if v]
            if bad:
                raise RuntimeError("sharded context not created on every rank - no rank enters ncclCommInitRank: " + "; ".join(bad))
            ids = [Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            self.ctx.comm_init(ids[0], rank, world)
            self.shard = None
        else:
            self.shard = DeviceShard(common_cfg, rank, world, device_index=common_cfg.get("device", 0))
            self.comm = TorchComm(dist, rank)
            self.ctx = self.shard.ctx

    def set_state(self, theta, mu, a_mat=None):
        if self.c_side:
            self.ctx.set_theta(theta)
            self.ctx.set_prior(mu)
            self.ctx.set_a_mat(theta if a_mat is None else a_mat)
        else:
            self.shard.set_state(theta, mu, a_mat)

    def tick(self, state, n_iters, eps=None, want_outputs=False, params=None):
        if self.c_side:
            return self.ctx.svmpc_tick(state, n_iters, eps, params, want_outputs=want_outputs)
        return tick((self.shard,), self.comm, state, n_iters, eps, params, want_outputs)

    def sync(self):
        if self.c_side:
            self.ctx.sync()
        else:
            self.shard.sync()
