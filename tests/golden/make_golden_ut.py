#!/usr/bin/env python3
"""Golden vectors for the unscented-transform rollouts ("DISCO" case), produced by running the REFERENCE.

TEST INFRASTRUCTURE, build container only:  python tests/golden/make_golden_ut.py   (writes tests/golden/disco_ut.npz)

Calls dust.controllers.disco.MultiDISCO(params_sampling=MerweScaledUTF(n=2, alpha=0.5)).forward twice - with the
controller's own noise (recorded) and with external actions - and `step("average")`, as demo/pendulum_example.py's DISCO
case does (pendulum_example.py:238-261 with demo/pendulum_config.yaml's utf block).  Nothing here re-implements the math.
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the shim and the RNG recorder)

import torch  # noqa: E402
import torch.distributions as dist  # noqa: E402
from dust.controllers.disco import MultiDISCO  # noqa: E402
from dust.models.pendulum import PendulumModel  # noqa: E402
from dust.utils.utf import MerweScaledUTF  # noqa: E402


def main():
    torch.manual_seed(11)
    N, H, S = 3, 7, 16  # H is NOT a multiple of the 5 sigma points: the (sigma, step) weight pattern of disco.py:314-316 shows
    tf = MerweScaledUTF(n=2, alpha=0.5)
    dyn = dist.Independent(dist.Uniform(torch.tensor([0.6, 0.7]), torch.tensor([1.3, 1.2])), 1)
    model = PendulumModel(length=dyn.mean[0], mass=dyn.mean[1], uncertain_params=("length", "mass"))
    ctrl = MultiDISCO(model.observation_space, model.action_space, H, N, S, temperature=0.8, a_cov=1.2 ** 2 * torch.eye(1),
                      inst_cost_fn=mg.pend_inst_cost, term_cost_fn=mg.pend_term_cost, params_sampling=tf, params_log_space=False)
    ctrl.a_mat = torch.randn(N, H, 1)
    state = torch.tensor([2.5, -0.5])
    g = dict(N=N, H=H, S=S, sigma_a=1.2, temperature=0.8, a_mat0=mg.npf(ctrl.a_mat), state=mg.npf(state),
             loc_weights=mg.npf(tf.loc_weights), dyn_mean=mg.npf(dyn.mean), dyn_var=mg.npf(dyn.variance),
             sigma_points=mg.npf(tf.compute_sigma_points(dyn.mean, dyn.variance.diag())))
    mg._REC.clear()
    costs, states, actions, omega, plp = ctrl.forward(state, model, dyn)
    z = [r for r in mg._REC if tuple(r.shape) == (S, N, H, 1)]
    assert len(z) == 1
    g.update(z=mg.npf(z[0]), costs=mg.npf(costs), actions=mg.npf(actions), omega=mg.npf(omega), a_mat1=mg.npf(ctrl.a_mat),
             a_mix=mg.npf(ctrl.a_mix), params_log_p=mg.npf(plp))
    c2 = copy.deepcopy(ctrl)
    out = c2.step(strategy="average", steps=1)
    g.update(step_average_actions=mg.npf(out), step_average_a_mat=mg.npf(c2.a_mat))
    ext = ctrl.a_mat.detach().clone().unsqueeze(0) + 0.7 * torch.randn(S, N, H, 1)
    costs2, _, _, omega2, _ = ctrl.forward(state, model, dyn, ext_actions=ext)
    g.update(ext_actions=mg.npf(ext), costs_ext=mg.npf(costs2), omega_ext=mg.npf(omega2), a_mat2=mg.npf(ctrl.a_mat))
    np.savez_compressed(os.path.join(mg.OUT, "disco_ut.npz"), **g)
    print("wrote disco_ut", {k: np.asarray(v).shape for k, v in g.items()})


if __name__ == "__main__":
    main()
