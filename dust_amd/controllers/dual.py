"""`DualSVMPC`: dual Stein-variational MPC as ONE object - the control-side SVGD (`SVMPC`) and the dynamics-side SVGD (`MPF`) composed
the way the reference composes them by hand in its simulation loop (dust/utils/simulations.py:104-138; demo/pendulum_example.py
"DuSt-MPC" case).  BASELINE.json's north_star names this surface ("DualSVMPC / SVMPC step() and forward()"); the reference has no
such class, so the names below are new and the SEMANTICS are the loop's:

    forward(state)            simulations.py:108-123   svmpc.optimize(state, dyn_dist); [step >= warm_up:] svmpc.forward(state, dyn_dist)
                                                        -> (a_seq [H, da], p_weights [N]); zero action sequence while warming up
    step(action, new_state)   simulations.py:132-138   mpf.optimize(action, new_state, bw, n_steps) -> (grad_norms, bw); the
                                                        controller's dynamics samples are drawn from the filter's refreshed prior
                                                        (dyn_dist = mpf.prior, simulations.py:79) from the next forward() on

Both halves run on the MI355X through the C ABI (dust_svmpc_tick / dust_svmpc_optimize + forward; dust_mpf_optimize;
dust_mpf_prior_sample).  `serve=True` turns on closed-loop serving for the control half when its shape allows it (nominal dynamics only:
a filter-coupled controller samples dynamics parameters per tick, which serving does not take - then it is a no-op)."""
import copy

import torch


class DualSVMPC:
    def __init__(self, svmpc, mpf=None, dyn_dist=None, mpf_bw=None, mpf_steps=20, warm_up=0, n_steps=None):
        """svmpc: dust_amd.inference.SVMPC; mpf: dust_amd.inference.MPF or None (control half only: dyn_dist, possibly None, is then
        the fixed distribution the controller samples dynamics parameters from); mpf_bw: bandwidth handed to mpf.optimize (None:
        silvermans_rule of the filter's particles, mpf.py:68-73); n_steps: SVGD iterations per control tick (None: svmpc.n_steps)."""
        self.svmpc, self.mpf = svmpc, mpf
        self.dyn_dist = mpf.prior if mpf is not None else dyn_dist
        self.mpf_bw, self.mpf_steps = mpf_bw, int(mpf_steps)
        self.warm_up, self.n_steps = int(warm_up), n_steps
        self.ticks = 0
        self.last_bw = None

    def __deepcopy__(self, memo):  # simulations.py:62,78: the loop deep-copies controller and filter per episode
        new = copy.copy(self)
        memo[id(self)] = new
        new.svmpc = copy.deepcopy(self.svmpc, memo)
        new.mpf = copy.deepcopy(self.mpf, memo)
        new.dyn_dist = new.mpf.prior if new.mpf is not None else self.dyn_dist
        return new

    # ---- attributes the loop reads (simulations.py:122, 140-160)
    @property
    def theta(self):
        return self.svmpc.theta

    @property
    def dyn_particles(self):
        return None if self.mpf is None else self.mpf.x

    @property
    def controller(self):
        return self.svmpc.likelihood.controller

    # ---- the control half of a tick (simulations.py:108-123)
    def forward(self, state):
        sv = self.svmpc
        sv.optimize(state, self.dyn_dist, n_steps=self.n_steps)
        self.ticks += 1
        if self.ticks <= self.warm_up:  # the loop applies a zero action while the particles warm up and does NOT roll them
            return torch.zeros(self.controller.hz_len, self.controller.dim_a), None
        return sv.forward(state, self.dyn_dist)

    # ---- the dynamics half (simulations.py:132-138)
    def step(self, action, new_state):
        if self.mpf is None:
            return None, None
        a = torch.as_tensor(action, dtype=torch.float).reshape(-1)
        grads, bw = self.mpf.optimize(a.squeeze() if a.numel() == 1 else a, new_state, bw=self.mpf_bw, n_steps=self.mpf_steps)
        self.last_bw = bw
        return grads, bw

    def tick(self, state, plant):
        """One full loop iteration with a host plant callable `plant(state, action) -> new_state`: forward, plant, step.
        Returns (action, new_state, p_weights)."""
        a_seq, pw = self.forward(state)
        action = a_seq[0]
        new_state = plant(state, action)
        self.step(action, new_state)
        return action, new_state, pw
