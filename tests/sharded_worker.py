"""TEST INFRASTRUCTURE: one rank of a multi-PROCESS sharded run on ONE GPU (tests/test_gpu_sharding_world2.py).
    python tests/sharded_worker.py <rank> <world> <dir> <case.json>
Runs the product's ShardedSVMPC (c_side=True: dust_comm_init + the C-side sharded tick) with tests/fake_rccl as the collective library
(DUST_RCCL_LIB) and a file-based stand-in for the two torch.distributed object collectives ShardedSVMPC uses to agree on the
communicator id; writes the tick outputs and the final particles to <dir>/out_<rank>.npz."""
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class FileDist:
    """all_gather_object / broadcast_object_list over files in a shared directory (what torch.distributed's gloo group does for
    ShardedSVMPC in a real launch)."""

    def __init__(self, d, rank, world):
        self.d, self.rank, self.world, self.seq = d, rank, world, 0

    def _put(self, name, obj):
        tmp = os.path.join(self.d, name + ".tmp%d" % self.rank)
        with open(tmp, "wb") as fh:
            pickle.dump(obj, fh)
        os.replace(tmp, os.path.join(self.d, name))

    def _get(self, name):
        p = os.path.join(self.d, name)
        t0 = time.time()
        while not os.path.exists(p):
            time.sleep(0.002)
            if time.time() - t0 > 120:
                raise TimeoutError(name)
        with open(p, "rb") as fh:
            return pickle.load(fh)

    def all_gather_object(self, out, obj):
        self.seq += 1
        self._put("ag%d_%d" % (self.seq, self.rank), obj)
        for r in range(self.world):
            out[r] = self._get("ag%d_%d" % (self.seq, r))

    def broadcast_object_list(self, lst, src=0):
        self.seq += 1
        if self.rank == src:
            self._put("bc%d" % self.seq, list(lst))
        got = self._get("bc%d" % self.seq)
        for i, v in enumerate(got):
            lst[i] = v


def case_inputs(case):
    from test_gpu_parity import _synthetic_case

    da, rng, mu, th, state, up, grid = _synthetic_case(case["model"], case["N"], case["S"], case["M"], case["H"])
    K, T = case["K"], case["T"]
    eps = rng.standard_normal((T, K, case["S"], case["N"], case["H"], da)).astype(np.float32) if case["ext_noise"] else None
    params = None if case["M"] == 1 else (1.0 + 0.1 * rng.standard_normal((T, K, case["M"], 1))).astype(np.float32)
    kw = dict(model=case["model"], N=case["N"], S=case["S"], M=case["M"], H=case["H"], kernel=case.get("kernel", "K1"), lr=0.5, sigma_a=1.0,
              sigma_p=1.0, uncertain_params=up, grid=grid, seed=11, optimizer=case.get("optimizer", "SGD"))
    return kw, mu, th, state, eps, params


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    case = json.load(open(sys.argv[4]))
    from dust_amd.parallel import ShardedSVMPC

    kw, mu, th, state, eps, params = case_inputs(case)
    sh = ShardedSVMPC(kw, rank, world, FileDist(d, rank, world), c_side=True)
    sh.set_state(th, mu)
    outs = []
    for t in range(case["T"]):
        a_seq, pw = sh.tick(state, case["K"], None if eps is None else eps[t], want_outputs=True, params=None if params is None else params[t])
        outs.append((a_seq.copy(), pw.copy()))
    sh.sync()
    np.savez(os.path.join(d, "out_%d.npz" % rank), theta=sh.ctx.get_theta(), a_mat=sh.ctx.get_a_mat(),
             a_seq=np.stack([o[0] for o in outs]), pw=np.stack([o[1] for o in outs]))
    sh.ctx.close()


if __name__ == "__main__":
    main()
