#!/usr/bin/env python3
"""f.1 golden: the REFERENCE's own closed-loop driver, `dust.utils.simulations.run_pendulum_simulation` (simulations.py:13-190),
run for three control ticks in the dual-inference configuration of demo/pendulum_example.py (SVMPC + MPF), with every random draw
and every per-tick product recorded.  TEST INFRASTRUCTURE - needs /root/reference:   python tests/golden/make_golden_driver.py

The reference steps gym's `Pendulum-v0` (simulations.py:49-53,129-130); gym is not installable here, so a stand-in `gym` module is
put in sys.modules whose env is the reference's own PendulumModel with g = 10 and the episode's (length, mass) - the plant the
build's driver uses too (dust_amd/utils/simulations.py).  The call ORDER per tick - optimize, (forward unless warming up), plant
step, mpf.optimize, cost - is the reference's code, untouched.  Its result frame hard-codes 200 rows (simulations.py:171-190), so with
steps = 3 the frame construction raises after the loop; everything the fixture holds was recorded inside the loop."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the shim, the RNG recorder, imports torch and the reference)

import torch  # noqa: E402
import torch.distributions as dist  # noqa: E402
from dust.models.pendulum import PendulumModel  # noqa: E402


class _Env:
    """Stand-in for gym's Pendulum-v0: the reference's PendulumModel with gym's g = 10 and the episode's length / mass."""

    def __init__(self):
        self.unwrapped = self
        self.l, self.m, self.state = 1.0, 1.0, None

    def reset(self):
        return None

    def step(self, action):
        plant = PendulumModel(g=10.0, length=float(self.l), mass=float(self.m))
        st = torch.as_tensor(self.state, dtype=torch.float).reshape(1, -1)
        a = torch.as_tensor(action, dtype=torch.float).reshape(1, -1).clamp(-2.0, 2.0)
        self.state = plant.step(st, a).reshape(-1)
        return self.state, 0.0, False, {}

    def render(self):
        pass

    def close(self):
        pass


gym = types.ModuleType("gym")
gym.make = lambda name: _Env()
sys.modules["gym"] = gym
from dust.utils import simulations as refsim  # noqa: E402
from dust.controllers.disco import MultiDISCO  # noqa: E402
from dust.inference.likelihoods import GaussianLikelihood  # noqa: E402
from dust.inference.mpf import MPF  # noqa: E402
from dust.inference.svgd import get_gmm  # noqa: E402
from dust.inference.svmpc import SVMPC  # noqa: E402

REC = dict(eps=[], params=[], theta_opt=[], theta_fwd=[], p_weights=[], a_seq=[], mpf_x=[], mpf_gn=[], state_in=[], mpf_bw_used=[])


class RecSVMPC(SVMPC):
    def optimize(self, state, params_dist, *a, **k):
        mg._REC.clear()
        REC["state_in"].append(mg.npf(state.reshape(-1)))
        out = super().optimize(state, params_dist, *a, **k)
        S, N = self.likelihood.n_samples, self.n_particles
        eps = [r for r in mg._REC if r.dim() == 4 and r.shape[0] == S and r.shape[1] == N]
        assert len(eps) == 1
        REC["eps"].append(mg.npf(eps[0]))
        REC["theta_opt"].append(mg.npf(self.theta))
        return out

    def forward(self, state, params_dist, *a, **k):
        a_seq, pw = super().forward(state, params_dist, *a, **k)
        REC["a_seq"].append(mg.npf(a_seq))
        REC["p_weights"].append(mg.npf(pw))
        REC["theta_fwd"].append(mg.npf(self.theta))
        return a_seq, pw


class RecMPF(MPF):
    def optimize(self, action, new_obs, *a, **k):
        g, bw = super().optimize(action, new_obs, *a, **k)
        REC["mpf_x"].append(mg.npf(self.x))
        REC["mpf_gn"].append(mg.npf(g))
        REC["mpf_bw_used"].append(np.float32(bw))
        return g, bw


_orig_mix_sample = dist.MixtureSameFamily.sample


def _rec_mix_sample(self, sample_shape=torch.Size()):
    out = _orig_mix_sample(self, sample_shape)
    if len(sample_shape) == 1 and out.dim() == 2:  # the controller's params_dist.sample([M]) (disco.py:171)
        REC["params"].append(mg.npf(out))
    return out


if __name__ == "__main__":
    torch.manual_seed(31)
    N, H, S, M, Mp = 6, 8, 8, 4, 10
    MPF_STEPS, MPF_BW, STEPS, WARM = 5, 0.2, 3, 1
    # `python tests/golden/make_golden_driver.py bwnull`: demo/pendulum_config.yaml's own setting, `mpf_bandwidth: null` - the filter's first
    # prior from bw_silverman of its particles (mpf.py:31-36) and every mpf.optimize with bw = silvermans_rule of the pooled particles
    # (mpf.py:68-73; KDEpy restated in oracle/ref_shim.py: third-party, parity unpinned) -> driver_pend_dual_bwnull.npz
    BWNULL = len(sys.argv) > 1 and sys.argv[1] == "bwnull"
    if BWNULL:
        MPF_BW = None
    env_model = PendulumModel()
    init_state = torch.tensor([3.0, 0.0])
    policies_prior = get_gmm(torch.randn(N, H, 1), torch.ones(N), 2.0 ** 2 * torch.eye(1))
    init_policies = policies_prior.sample([N])
    init_policies0 = init_policies.detach().clone()  # (SVMPC takes the tensor itself as its particles and SGD updates it in place)
    dynamics_prior = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.6]), torch.tensor([1.3, 1.3])), 1)
    controller = MultiDISCO(observation_space=env_model.observation_space, action_space=env_model.action_space, hz_len=H, action_samples=S,
                            params_samples=M, temperature=1.0, a_cov=2.0 ** 2 * torch.eye(1), inst_cost_fn=mg.pend_inst_cost,
                            term_cost_fn=mg.pend_term_cost, params_sampling=True, n_policies=N, params_log_space=False)
    svmpc_kwargs = dict(init_particles=init_policies, prior=policies_prior, kernel=mg.ref_shim.RBFKernel(), n_particles=N, bw_scale=1.0,
                        n_steps=1, optimizer_class=torch.optim.SGD, lr=2.0)
    mpf_init = dynamics_prior.sample([Mp])
    lik = GaussianLikelihood(initial_obs=init_state, obs_std=0.1, model=PendulumModel(uncertain_params=("length", "mass")), log_space=False)
    mpf = RecMPF(init_particles=mpf_init.clone(), likelihood=lik, optimizer_class=torch.optim.SGD, lr=1e-3, bw=MPF_BW, bw_scale=1.0)
    truth = [{"length": torch.tensor(0.9), "mass": torch.tensor(1.1)}]
    refsim.SVMPC = RecSVMPC
    dist.MixtureSameFamily.sample = _rec_mix_sample
    try:
        refsim.run_pendulum_simulation(init_state=init_state, init_policies=init_policies, model_kwargs={"uncertain_params": ("length", "mass")},
                                       dyn_dist=dynamics_prior, experiment_params=truth, controller=controller, use_exact_model=False,
                                       use_svmpc=True, svmpc_kwargs=svmpc_kwargs, lik_kwargs={"alpha": 1.0, "n_samples": S}, mpf=mpf,
                                       mpf_bw=MPF_BW, mpf_steps=MPF_STEPS, episodes=1, steps=STEPS, render=False, warm_up=WARM)
        raise SystemExit("the reference's 200-row frame accepted %d steps?" % STEPS)
    except ValueError as e:  # simulations.py:171-190: arrays of length `steps` against an index of 200
        print("(expected) frame construction failed after the loop:", str(e)[:80])
    finally:
        dist.MixtureSameFamily.sample = _orig_mix_sample
    assert len(REC["eps"]) == STEPS and len(REC["params"]) == STEPS and len(REC["mpf_x"]) == STEPS and len(REC["a_seq"]) == STEPS - WARM
    g = dict(N=N, H=H, S=S, M=M, Mp=Mp, mpf_steps=MPF_STEPS, mpf_bw=(-1.0 if MPF_BW is None else MPF_BW), steps=STEPS, warm_up=WARM, sigma=2.0, lr=2.0, mpf_lr=1e-3,
             obs_std=0.1, init_state=mg.npf(init_state), mu0=mg.npf(policies_prior.component_distribution.base_dist.loc),
             init_policies=mg.npf(init_policies0), mpf_init=mg.npf(mpf_init), true_length=0.9, true_mass=1.1)
    for k, v in REC.items():
        g[k] = np.stack(v)
    name = "driver_pend_dual_bwnull" if BWNULL else "driver_pend_dual"
    np.savez_compressed(os.path.join(mg.OUT, name + ".npz"), **g)
    print("wrote", name, {k: v.shape for k, v in g.items() if hasattr(v, "shape") and v.ndim > 1})
