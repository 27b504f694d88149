"""Cost-function families the HIP kernels implement, and recognition of user callables.

The reference takes arbitrary Python callables (`inst_cost_fn`, `term_cost_fn`; the pendulum ones live in the demo script,
demo/pendulum_example.py:21-28).  A GPU kernel cannot call back into Python and this backend has no CPU fallback, so a
callable is accepted only if it belongs to a known family: either it is one of the tagged callables below / a bound
`Particle.default_*_cost`, or - so that the demo's own `inst_cost` works unchanged - it is PROBED on a few states and
matches  w_cos (cos th - 1)^2 + w_vel thd^2  to 1e-6.  Anything else raises NotImplementedError."""
import torch


class PendulumQuadCos:
    """inst = w_cos (cos th - 1)^2 + w_vel thd^2 ; term = inst (demo/pendulum_example.py:21-28)."""

    family = "pendulum_quadcos"

    def __init__(self, w_cos=50.0, w_vel=1.0):
        self.w_cos, self.w_vel = float(w_cos), float(w_vel)

    def inst_cost(self, states, controls=None, n_pol=1, debug=None):
        th, thd = states.chunk(2, dim=1)
        return self.w_cos * (th.cos() - 1) ** 2 + self.w_vel * thd ** 2

    def term_cost(self, states, n_pol=1, debug=None):
        return self.inst_cost(states).squeeze()


class QuadraticCost:
    """Quadratic state / control cost for any model (the skid-steer family uses it: the reference ships no cost for that model and
    its MultiDISCO takes any callable with this signature, disco.py:294-346):
        inst(x, a) = sum_k w_state[k] (x_k - goal_k)^2 + sum_d w_ctrl[d] a_d^2        term(x) = sum_k w_term[k] (x_k - goal_k)^2
    `inst_cost` / `term_cost` are plain torch, so the same object drives the reference's controller and this one's kernels."""

    family = "quadratic"

    def __init__(self, goal, w_state, w_term=None, w_ctrl=None):
        self.goal = torch.as_tensor(goal, dtype=torch.float).reshape(-1)
        self.w_state = torch.as_tensor(w_state, dtype=torch.float).reshape(-1)
        self.w_term = self.w_state.clone() if w_term is None else torch.as_tensor(w_term, dtype=torch.float).reshape(-1)
        self.w_ctrl = None if w_ctrl is None else torch.as_tensor(w_ctrl, dtype=torch.float).reshape(-1)

    def inst_cost(self, states, controls=None, n_pol=1, debug=None):
        c = (((states - self.goal) ** 2) * self.w_state).sum(-1)
        if controls is not None and self.w_ctrl is not None:
            c = c + ((controls ** 2) * self.w_ctrl).sum(-1)
        return c

    def term_cost(self, states, n_pol=1, debug=None):
        return (((states - self.goal) ** 2) * self.w_term).sum(-1)


def _probe_quadcos(fn):
    pts = torch.tensor([[0.3, 0.0], [0.0, 1.7], [2.1, -0.4], [-1.2, 3.3], [5.9, -7.1], [3.14159, 0.5]])
    try:
        out = torch.as_tensor(fn(pts)).reshape(-1).double()
    except Exception:
        return None
    if out.numel() != pts.shape[0]:
        return None
    a = ((pts[:, 0].cos() - 1) ** 2).double()
    b = (pts[:, 1] ** 2).double()
    A = torch.stack([a, b], dim=1)
    sol = torch.linalg.lstsq(A, out.unsqueeze(1)).solution.reshape(-1)
    if torch.allclose(A @ sol, out, rtol=1e-6, atol=1e-6):
        snap = lambda v: float('%.6g' % float(v))  # probing in fp32 leaves ~1e-8 noise on the fitted weights
        return snap(sol[0]), snap(sol[1])
    return None


def recognise(model, inst_cost_fn, term_cost_fn):
    """-> dict of dust_config cost fields, or raise NotImplementedError."""
    fam = getattr(model, "family", None)
    if fam == "pendulum":
        owner = getattr(inst_cost_fn, "__self__", None)
        if isinstance(owner, PendulumQuadCos):
            w = (owner.w_cos, owner.w_vel)
        else:
            w = _probe_quadcos(inst_cost_fn) if inst_cost_fn is not None else None
        if w is None:
            raise NotImplementedError("inst_cost_fn is not of the form w_cos (cos th - 1)^2 + w_vel thd^2: no HIP kernel for it "
                                      "(dust_amd has no CPU fallback)")
        wt = _probe_quadcos(lambda s: term_cost_fn(s)) if term_cost_fn is not None else None
        if wt is None or abs(wt[0] - w[0]) > 1e-6 * max(1, abs(w[0])) or abs(wt[1] - w[1]) > 1e-6 * max(1, abs(w[1])):
            raise NotImplementedError("term_cost_fn must equal the instantaneous pendulum cost (demo/pendulum_example.py:27-28)")
        return dict(w_cos=w[0], w_vel=w[1])
    if fam == "particle":
        ok = (getattr(inst_cost_fn, "__func__", None) is type(model).default_inst_cost and
              getattr(term_cost_fn, "__func__", None) is type(model).default_term_cost)
        if not ok:
            raise NotImplementedError("Particle needs model.default_inst_cost / model.default_term_cost (no CPU fallback)")
        m = inst_cost_fn.__self__
        return dict(target=tuple(float(v) for v in m.target), w_state=tuple(float(v) for v in m.w_state),
                    w_term=tuple(float(v) for v in m.w_term), w_ctrl=tuple(float(v) for v in m.w_ctrl), w_obs=float(m.w_obs))
    if fam == "skid_steer":
        owner = getattr(inst_cost_fn, "__self__", None)
        if not isinstance(owner, QuadraticCost) or getattr(term_cost_fn, "__self__", None) is not owner:
            raise NotImplementedError("SkidSteerRobot runs with dust_amd.costs.QuadraticCost(...).inst_cost / .term_cost (the reference ships "
                                      "no cost for this model; an opaque callable cannot run on the device and there is no CPU fallback)")
        if owner.goal.numel() != 5 or owner.w_state.numel() != 5 or owner.w_term.numel() != 5:
            raise ValueError("QuadraticCost for SkidSteerRobot needs 5 state entries (x, y, theta, v, omega)")
        wc = owner.w_ctrl if owner.w_ctrl is not None else torch.zeros(2)
        return dict(goal=tuple(float(v) for v in owner.goal), w_quad_state=tuple(float(v) for v in owner.w_state),
                    w_quad_term=tuple(float(v) for v in owner.w_term), w_quad_ctrl=tuple(float(v) for v in wc))
    raise NotImplementedError("model family %r has no HIP kernel" % (fam,))
