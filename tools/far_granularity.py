"""How fine must the far test be?  The cfg4 set of bench.py after `age` ticks: share of blocks with NO near pair (K1: G <= T, prior: G <= T + 2 (lm_j - m0_i)) at several
block shapes (queries x keys).  Host arithmetic (numpy, blockwise).  python tools/far_granularity.py [age]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from dust_amd import Context

age = int(sys.argv[1]) if len(sys.argv) > 1 else 140
c4 = bench.CFG4
mu4, theta4 = bench.synth(c4["N"], c4["H"], 2, spread=1.0)
one = Context(model="particle", N=c4["N"], S=c4["S"], M=c4["M"], H=c4["H"], kernel="K1", lr=100.0, alpha=1.0, sigma_a=1.0, sigma_p=1.0,
              uncertain_params=("mass",), grid=bench.particle_grid(), device=0, seed=1234)
one.set_theta(theta4); one.set_prior(mu4); one.set_a_mat(theta4)
p4 = (1.0 + 0.1 * np.random.default_rng(5).standard_normal((c4["n_iters"], c4["M"], 1))).astype(np.float32)
st4 = np.array([-9.0, -9.0, 0.0, 0.0], np.float32)
for k in range(age - 1):
    one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=False)
a_seq, pw = one.svmpc_tick(st4, c4["n_iters"], params=p4, want_outputs=True)
X = one.get_theta().reshape(c4["N"], -1).astype(np.float64)
one.close()
N = X.shape[0]
with np.errstate(divide="ignore"):
    lm = np.log(pw.astype(np.float64))
print("age %d: weights max %.3g, #(w > 1e-6) %d, log w range [%.1f, %.1f]" % (age, pw.max(), int((pw > 1e-6).sum()), lm[np.isfinite(lm)].min(), lm.max()))
T = 60.0
nrm = (X * X).sum(1)
# m0: own logit and the logits against the heaviest particle of each 64-key chunk
cand = np.array([c0 + int(np.argmax(lm[c0:c0 + 64])) for c0 in range(0, N, 64)])
dc = nrm[:, None] + nrm[cand][None, :] - 2.0 * X @ X[cand].T
m0 = np.maximum(lm, (lm[cand][None, :] - 0.5 * dc).max(1))
near = np.zeros((N, N), bool)
B = 2048
d2hist = np.zeros(8, np.int64)
for i0 in range(0, N, B):
    G = nrm[i0:i0 + B, None] + nrm[None, :] - 2.0 * X[i0:i0 + B] @ X.T
    thr = T + 2.0 * np.maximum(0.0, lm[None, :] - m0[i0:i0 + B, None])
    thr = np.where(np.isfinite(thr), thr, np.inf)
    near[i0:i0 + B] = G <= thr
    d2hist += np.histogram(G, bins=[-1, 10, 30, 60, 100, 200, 400, 800, 1e9])[0]
print("pair distances G: <10 %d, <30 %d, <60 %d, <100 %d, <200 %d, <400 %d, <800 %d, more %d" % tuple(d2hist))
print("near pairs: %.3g of all (K1-only G <= 60 share would be %.3g)" % (near.mean(), d2hist[:3].sum() / float(N) ** 2))
for q, k in ((96, 64), (64, 64), (16, 64), (2, 64), (1, 64), (16, 16), (64, 16), (1, 16)):
    nq, nk = N // q * q, N // k * k
    blk = near[:nq, :nk].reshape(nq // q, q, nk // k, k).any(axis=(1, 3))
    print("  blocks %3d queries x %2d keys: %.4f without a near pair" % (q, k, 1.0 - blk.mean()))

# --- round 6: how long are a query tile's near-key lists (pairwise_packed.hpp), and what would ordering the queries by cluster buy? ---
def list_stats(order, name):
    tiles = [order[t:t + 96] for t in range(0, N, 96)]
    L = np.array([int(near[t].any(axis=0).sum()) for t in tiles])          # merged list: keys near to SOME query of the tile
    qshare = []
    for t in tiles[::8]:
        keys = np.flatnonzero(near[t].any(axis=0))
        for u0 in range(0, len(keys), 64):
            qshare.append(float(near[np.ix_(t, keys[u0:u0 + 64])].any(axis=1).mean()))
    units = int(np.ceil(L / 64.0).sum())
    print("  %-34s list length per tile: mean %.0f, max %d; packed units %d; near queries per unit %.2f" % (name, L.mean(), L.max(), units, float(np.mean(qshare))))

deg = near.sum(1)
print("near keys per query: mean %.1f, median %d, p90 %d, max %d" % (deg.mean(), int(np.median(deg)), int(np.percentile(deg, 90)), int(deg.max())))
list_stats(np.arange(N), "index order")
first_chunk = np.array([int(np.flatnonzero(near[i])[0]) // 64 for i in range(N)])
list_stats(np.argsort(first_chunk, kind="stable"), "by first near chunk")
first_key = np.array([int(np.flatnonzero(near[i])[0]) for i in range(N)])
list_stats(np.argsort(first_key, kind="stable"), "by first near key (leader)")
# the same from the previous tick's structure is what the device could use; label propagation = connected components of the near graph
lab = np.arange(N)
for it in range(20):
    new = np.array([lab[np.flatnonzero(near[i])].min() for i in range(N)])
    if (new == lab).all():
        break
    lab = new
sizes = np.bincount(lab)
sizes = sizes[sizes > 0]
print("connected components of the near graph: %d (largest %d, singletons %d)" % (len(sizes), sizes.max(), int((sizes == 1).sum())))
list_stats(np.argsort(lab, kind="stable"), "by connected component")
