#!/usr/bin/env python3
"""Round-2 golden vectors, produced by running the REFERENCE (lubaroli/dust) in this container - TEST INFRASTRUCTURE.

  python tests/golden/make_golden_r2.py          (needs /root/reference; writes tests/golden/*.npz)

  * pend_k1_adam / part_k1_adam: SVMPC with the reference's class-default optimiser, torch.optim.Adam (svgd.py:115), over three
    control ticks: pins the Adam step AND the reset of its state at every forward() (SVMPC.roll makes a new parameter tensor,
    svmpc.py:142-158, so torch's per-tensor optimiser state starts again);
  * pend_k1_f64 / pend_k1_mid_f64: the K1 kernel branch (svmpc.py:76-83; gpytorch RBFKernel semantics - third party, absent) ALSO
    evaluated in float64 on the recorded fp32 inputs (theta, score): the fp32 stand-in carries cancellation noise of its
    matmul-trick distance, the float64 run does not, so oracle and HIP can be held to 1e-5 against it.
  * mpf_pend_adam / mpf_part_log_adam: MPF with the class-default optimiser (Adam; its state persists across optimize() calls:
    the optimiser is built once in MPF.__init__, mpf.py:24): two filter updates each.
  * pend_k2_fixedbw / part_k2shared_fixedbw: iid_mp(RBF(bandwidth >= 0)) - the fixed-bandwidth branch of base_kernels.py:66-67.
  * skid_steer: SkidSteerRobot.step (skid_steer_robot.py:73-122) on a handful of states, with default and per-row parameters.
Same recording machinery as make_golden.py (run_svmpc, run_mpf); nothing here re-implements the reference's arithmetic.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (installs the shim, imports the reference)

if __name__ == "__main__":
    mg.run_svmpc("pend_k1_adam", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K1", n_iters=3, n_ticks=3, seed=20, optimizer="Adam",
                 lr_override=0.1)
    mg.run_svmpc("part_k1_adam", "particle", N=8, H=12, S=8, M=4, kernel_kind="K1", weighted_prior=True, n_iters=2, n_ticks=3, seed=21,
                 params_kind="logmass_gmm", optimizer="Adam", lr_override=0.5)
    mg.run_svmpc("pend_k1_f64", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K1", n_iters=2, n_ticks=2, seed=22, theta_shrink=0.05,
                 k1_f64=True, lr_override=0.02)  # (small step: the particles stay within a few lengthscales of each other)
    mg.run_svmpc("pend_k1_mid_f64", "pendulum", N=24, H=12, S=8, M=1, kernel_kind="K1", n_iters=2, n_ticks=1, seed=23, theta_shrink=0.12,
                 k1_f64=True, lr_override=0.02)
    mg.run_mpf("mpf_pend_adam", "pendulum", Mp=10, n_steps=6, log_space=False, bw=0.08, optimizer="Adam", lr_override=0.01)
    mg.run_mpf("mpf_part_log_adam", "particle", Mp=12, n_steps=12, log_space=True, bw=0.5, optimizer="Adam", lr_override=0.02)
    mg.run_svmpc("pend_k2_fixedbw", "pendulum", N=16, H=10, S=8, M=1, kernel_kind="K2", k2_bandwidth=0.7, n_iters=2, n_ticks=2, seed=24)
    mg.run_svmpc("part_k2shared_fixedbw", "particle", N=8, H=12, S=8, M=4, kernel_kind="K2shared", k2_bandwidth=1.5, n_iters=2, n_ticks=1,
                 seed=25, params_kind="logmass_gmm")
    # plant-side skid-steer model: step() only - the reference has no cost family / demo for it
    import numpy as np
    import torch
    from dust.models.skid_steer_robot import SkidSteerRobot

    torch.manual_seed(31)
    m = SkidSteerRobot(delta_t=0.1)
    states = torch.randn(7, 5)
    actions = torch.randn(7, 2) * 0.6  # some beyond the +-0.5 wheel-speed bounds
    nxt = m.step(states, actions, None)
    pd = {"x_icr": torch.rand(7, 1) * 0.3 + 0.1, "wheel_radius": torch.rand(7, 1) * 0.05 + 0.04, "axial_distance": torch.rand(7, 1) * 0.2 + 0.4}
    nxt_p = m.step(states, actions, pd)
    m2 = SkidSteerRobot(delta_t=0.05, x_icr=0.1, wheel_radius=0.08, axial_distance=0.5, min_wheel_speed=-1.0, max_wheel_speed=0.8)
    nxt2 = m2.step(states, actions * 2, None)
    np.savez_compressed(os.path.join(mg.OUT, "skid_steer.npz"), states=states.numpy(), actions=actions.numpy(), next=nxt.numpy(),
                        x_icr=pd["x_icr"].numpy(), wheel_radius=pd["wheel_radius"].numpy(), axial_distance=pd["axial_distance"].numpy(),
                        next_params=nxt_p.numpy(), next2=nxt2.numpy())
