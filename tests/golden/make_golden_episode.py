#!/usr/bin/env python3
"""f.1 golden: the REFERENCE's own particle episode driver, `dust.utils.simulations.run_particle_episode`
(simulations.py:197-260; the loop demo/particle_example.py:150-254 runs inline), on the 2-D point mass with the obstacle grid:
the mass-change event at steps // 4 (simulations.py:220-221), zero actions while warming up, crash termination (cost inf,
235-242) and goal termination (within 1.0 of the target, 243-244).  Every random draw and every per-tick product is recorded.
TEST INFRASTRUCTURE - needs /root/reference:   python tests/golden/make_golden_episode.py

Three episodes from one configuration (N = 6 policies, H = 8, S = 8 action samples, M = 4 log-mass samples from a GMM):
  run    8 steps from the demo's start state: load + 1.5 at step 2, no termination
  goal   starts 1.02 from the target drifting towards it, no warm-up: ends as soon as |target - state| <= 1
  crash  starts one cell from an obstacle at full speed towards it: ends with cost inf"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the shim, the RNG recorder, imports torch and the reference)
import make_golden_driver as mgd  # noqa: E402  (gym stand-in, recording SVMPC, recorder of the controller's dynamics samples)

import torch  # noqa: E402
import torch.distributions as dist  # noqa: E402
from dust.controllers.disco import MultiDISCO  # noqa: E402
from dust.inference.likelihoods import ExponentiatedUtility  # noqa: E402
from dust.inference.svgd import get_gmm  # noqa: E402
from dust.models.particle import Particle  # noqa: E402
from dust.utils import simulations as refsim  # noqa: E402

N, H, S, M = 6, 8, 8, 4
SIGMA, LR, LOAD, WARM = 5.0, 100.0, 1.5, 1


def episode(tag, init_state, steps, seed, warm=WARM):
    torch.manual_seed(seed)
    for v in mgd.REC.values():
        v.clear()
    model = Particle(**mg.PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    mu0 = torch.randn(N, H, 2)
    prior = get_gmm(mu0, torch.ones(N), SIGMA ** 2 * torch.eye(2))
    init_policies = prior.sample([N])
    init_policies0 = init_policies.detach().clone()
    # an MPF-style prior over the log-mass (particle_config.yaml:24-35): GMM over 16 particles, bandwidth 0.5
    x = dist.Normal(2.0, 0.1).sample([16, 1]).clamp(min=1e-6).log()
    dyn = dist.MixtureSameFamily(dist.Categorical(torch.ones(16)),
                                 dist.Independent(dist.MultivariateNormal(loc=x, covariance_matrix=0.5 ** 2 * torch.eye(1)), 0))
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=1.0, a_cov=SIGMA ** 2 * torch.eye(2),
                      params_sampling=True, params_samples=M, params_log_space=True, inst_cost_fn=model.default_inst_cost,
                      term_cost_fn=model.default_term_cost)
    lik = ExponentiatedUtility(1.0, controller=ctrl, model=model, n_samples=S)
    svmpc = mgd.RecSVMPC(init_particles=init_policies, prior=prior, likelihood=lik, kernel=mg.ref_shim.RBFKernel(), n_particles=N,
                         bw_scale=1.0, n_steps=1, optimizer_class=torch.optim.SGD, lr=LR, weighted_prior=True)
    state0 = torch.tensor(init_state, dtype=torch.float)
    dist.MixtureSameFamily.sample = mgd._rec_mix_sample
    try:
        cum = refsim.run_particle_episode(state0, model, dyn, ctrl, use_svmpc=True, warm_up=warm, svmpc=svmpc, load=LOAD, steps=steps)
    finally:
        dist.MixtureSameFamily.sample = mgd._orig_mix_sample
    n_run = len(mgd.REC["eps"])
    # the controller's dynamics samples: one [M, 1] draw per optimize (disco.py:171); prior.sample([N]) draws are [N, H, 2] - skipped
    params = [p for p in mgd.REC["params"] if p.shape == (M, 1)]
    assert len(params) == n_run, (len(params), n_run)
    g = dict(N=N, H=H, S=S, M=M, sigma=SIGMA, lr=LR, load=LOAD, warm_up=warm, steps=steps, steps_run=n_run, init_state=mg.npf(state0),
             mu0=mg.npf(mu0), init_policies=mg.npf(init_policies0), dyn_means=mg.npf(x), dyn_bw=0.5,
             cum_cost=np.float32(float(cum)), state_in=np.stack(mgd.REC["state_in"]), eps=np.stack(mgd.REC["eps"]), params=np.stack(params),
             theta_opt=np.stack(mgd.REC["theta_opt"]))
    if mgd.REC["a_seq"]:
        g.update(a_seq=np.stack(mgd.REC["a_seq"]), p_weights=np.stack(mgd.REC["p_weights"]), theta_fwd=np.stack(mgd.REC["theta_fwd"]))
    np.savez_compressed(os.path.join(mg.OUT, "episode_part_%s.npz" % tag), **g)
    print("wrote episode_part_%s: %d of %d steps, cum_cost %s" % (tag, n_run, steps, float(cum)))
    return g


def cell_next_to_obstacle(model):
    """A free cell whose +x neighbour is occupied, away from the map border: (x, y) of its centre."""
    m = model.obst_map.map.numpy() if hasattr(model.obst_map.map, "numpy") else np.asarray(model.obst_map.map)
    nx, ny = m.shape
    for ix in range(nx // 2, nx - 2):
        for iy in range(ny // 2, ny - 2):
            if m[ix, iy] == 0 and m[ix - 1, iy] == 0 and m[ix + 1, iy] == 1:
                cs = mg.PARTICLE_ENV["map_cell_size"]
                return (ix - nx / 2 + 0.5) * cs, (iy - ny / 2 + 0.5) * cs
    raise SystemExit("no free cell next to an obstacle?")


if __name__ == "__main__":
    r = episode("run", [-9.0, -9.0, 0.0, 0.0], steps=8, seed=41)
    assert r["steps_run"] == 8 and np.isfinite(r["cum_cost"])
    # a start just outside the goal radius, drifting in, no warm-up: optimize + forward, one plant step, then the goal test ends it
    gl = episode("goal", [8.37, 8.37, 0.25, 0.25], steps=8, seed=42, warm=0)
    assert gl["steps_run"] < 8 and np.isfinite(gl["cum_cost"]) and "a_seq" in gl, "goal episode did not terminate early"
    probe = Particle(**mg.PARTICLE_ENV, uncertain_params=["mass"], mass=torch.tensor(2.0))
    cx, cy = cell_next_to_obstacle(probe)
    cr = episode("crash", [cx, cy, 5.0, 0.0], steps=8, seed=43)
    assert not np.isfinite(cr["cum_cost"]) and cr["steps_run"] < 8, "crash episode did not crash"
