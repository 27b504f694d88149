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
dust_mpf_prior_sample).  `fused=True` (round 6) runs a whole control period - the filter update for the action just applied, Silverman's
bandwidth ON THE DEVICE when none is given, the controller's dynamics samples drawn from the refreshed filter prior on the device, the
control tick - in ONE C call (dust_dual_tick): step() then only notes (action, new_state) and the next forward() carries it out; the
filter's particles are current again after that forward().  The draws come from the library's Philox stream (not torch's), so a run
with recorded draws (`draw_source`) stays on the unfused path.  `serve=True` turns on closed-loop serving for the control half when its shape allows it (nominal dynamics only:
a filter-coupled controller samples dynamics parameters per tick, which serving does not take - then it is a no-op)."""
import copy

import torch


class DualSVMPC:
    def __init__(self, svmpc, mpf=None, dyn_dist=None, mpf_bw=None, mpf_steps=20, warm_up=0, n_steps=None, fused=False, seed=0):
        """svmpc: dust_amd.inference.SVMPC; mpf: dust_amd.inference.MPF or None (control half only: dyn_dist, possibly None, is then
        the fixed distribution the controller samples dynamics parameters from); mpf_bw: bandwidth handed to mpf.optimize (None:
        silvermans_rule of the filter's particles, mpf.py:68-73); n_steps: SVGD iterations per control tick (None: svmpc.n_steps)."""
        self.svmpc, self.mpf = svmpc, mpf
        self.dyn_dist = mpf.prior if mpf is not None else dyn_dist
        self.mpf_bw, self.mpf_steps = mpf_bw, int(mpf_steps)
        self.warm_up, self.n_steps = int(warm_up), n_steps
        self.ticks = 0
        self.last_bw = None
        self.fused, self._seed, self._pending = bool(fused), int(seed), None

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
        if self.mpf is None:
            return None
        self._flush()
        return self.mpf.x

    @property
    def controller(self):
        return self.svmpc.likelihood.controller

    # ---- the control half of a tick (simulations.py:108-123)
    def _can_fuse(self):
        ctrl = self.controller
        return (self.fused and self.mpf is not None and self.dyn_dist is self.mpf.prior and getattr(ctrl, "draw_source", None) is None
                and self.mpf.draw_source is None and self.svmpc.roll_strategy != "resample" and self.ticks >= self.warm_up)

    def forward(self, state):
        sv = self.svmpc
        if self._can_fuse():  # one C call: (pending filter update) -> dynamics samples on the device -> optimize + forward
            ctx = sv._ctx(self.dyn_dist)
            n_steps = sv.n_steps if self.n_steps is None else self.n_steps
            pend, self._pending = self._pending, None
            self._seed += 1
            a_prev = None if pend is None else pend[0]
            a_seq, pw, bw = ctx.dual_tick(self.mpf._dev, sv._state(state), a_prev, n_steps, self.mpf_steps, self.mpf_bw, self._seed)
            if pend is not None:
                self.last_bw = bw
            sv._prior_stale = True
            self.ticks += 1
            return torch.from_numpy(a_seq), torch.from_numpy(pw)
        self._flush()
        sv.optimize(state, self.dyn_dist, n_steps=self.n_steps)
        self.ticks += 1
        if self.ticks <= self.warm_up:  # the loop applies a zero action while the particles warm up and does NOT roll them
            return torch.zeros(self.controller.hz_len, self.controller.dim_a), None
        return sv.forward(state, self.dyn_dist)

    # ---- the dynamics half (simulations.py:132-138)
    def _flush(self):
        """A filter update noted by a fused step() and not carried out yet: run it now (the unfused path is about to read the filter)."""
        pend, self._pending = self._pending, None
        if pend is not None:
            a = torch.as_tensor(pend[0], dtype=torch.float).reshape(-1)
            _, self.last_bw = self.mpf.optimize(a.squeeze() if a.numel() == 1 else a, pend[1], bw=self.mpf_bw, n_steps=self.mpf_steps)

    def step(self, action, new_state):
        if self.mpf is None:
            return None, None
        if self._can_fuse():  # carried out by the next forward(new_state): one C call for the whole period
            self._flush()
            self._pending = (torch.as_tensor(action, dtype=torch.float).reshape(-1).numpy().copy(), torch.as_tensor(new_state, dtype=torch.float).reshape(-1).clone())
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
