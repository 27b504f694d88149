"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np


def relerr(a, b, floor=0.0):
    """max |a-b| / (max|b| + floor): tensor-level relative error, the form SURVEY section 3 quotes."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + floor + 1e-300))


def elem_relerr(a, b, atol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float((np.abs(a - b) / (np.abs(b) + atol + 1e-300)).max())


def elemerr(a, b, floor=None):
    """ELEMENT-WISE relative error with a stated absolute floor: max_i |a_i - b_i| / (|b_i| + floor).

    floor defaults to rms(b): an entry is held to `tol` relative to its own magnitude or to the tensor's RMS, whichever is
    larger (sums with cancellation - phi, grad_pri - cannot be relatively exact in their near-zero entries; the RMS floor is
    1-2 orders tighter than the tensor-max form `relerr` and cannot hide a wrong small entry behind one large one)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if floor is None:
        floor = float(np.sqrt(np.mean(b * b)))
    return float((np.abs(a - b) / (np.abs(b) + floor + 1e-300)).max())


def scenario_kwargs(g):
    """Oracle/ctx constructor kwargs for a golden SVMPC scenario (tests/golden/make_golden.py:run_svmpc)."""
    kind = str(g["model_kind"])
    kw = dict(model=kind, N=int(g["N"]), S=int(g["S"]), M=int(g["M"]), H=int(g["H"]))
    if kind == "pendulum":
        if "params" in g:
            kw["uncertain_params"] = ("length", "mass")
    else:
        kw["uncertain_params"] = ("mass",)
        kw["mass"] = 2.0
        kw["mass_0dim"] = True
        kw["params_log_space"] = bool(int(g["params_log_space"]))
        kw["params_scalar_event"] = bool(int(g["params_scalar_event"]))
        if "a_cov" in g:  # full 2 x 2 covariances (tests/golden/make_golden_r5.py run_svmpc_cov)
            kw["a_cov"], kw["p_cov"] = np.asarray(g["a_cov"], np.float32), np.asarray(g["p_cov"], np.float32)
        if "control_type" in g:  # round-5 fixtures (tests/golden/make_golden_r5.py): control noise / velocity control
            kw["control_type"] = str(g["control_type"])
            kw["noise_std"] = tuple(float(v) for v in g["dyn_std"])
            if kw["control_type"] == "velocity":  # a two-state model (particle.py:41-48, 307-322)
                kw.update(target=(4.0, 4.5), w_state=(0.5, 0.5), w_term=(1e3, 1e3))
    return kw


def ctx_kwargs(g):
    """scenario_kwargs plus what only the device context takes (the oracle reads the noise per call)."""
    kw = scenario_kwargs(g)
    if "deterministic" in g:
        kw["deterministic"] = bool(int(g["deterministic"]))
    return kw


def feed_ctrl_noise(c, g, t, k=None):
    """Hand the device context the recorded control-noise draws of tick t (all SVGD iterations, or iteration k alone): one set per
    rollout launch that follows (dust_set_ctrl_noise)."""
    if "ctrl_noise" in g:
        c.set_ctrl_noise(g["ctrl_noise"][t] if k is None else g["ctrl_noise"][t, k][None])


def ctrl_noise_of(g, t, k):
    """Recorded control-channel draws [H][M*S*N][da] of SVGD iteration k of tick t, or None (deterministic fixtures)."""
    return g["ctrl_noise"][t, k] if "ctrl_noise" in g else None


def is_adam(name):
    """Golden scenarios run with the reference's class-default optimiser (tests/golden/make_golden_r2.py)."""
    return name.endswith("_adam")


def tick2_ticks_expected(g, name, calls_per_tick=1):
    """How many of a golden scenario's optimize() / tick() calls the owner-computes one-launch kernel (dust_amd/csrc/tick2.hpp) serves:
    every call after the first forward() - from then on the prior means alias the particles (svgd.py:87) - when the shape is one the
    kernel takes (K1 / IMQ, N % 4 == 0, H * d_a <= 32, isotropic prior scale, no control cost).  The golden chain tests assert this
    count, so that which kernel produced the compared numbers is on record."""
    T = int(g["eps"].shape[0])
    N, D = int(g["N"]), int(g["H"]) * int(g["da"])
    sp = np.atleast_1d(np.asarray(g["sigma_p"], np.float64)).reshape(-1)
    # (velocity control and non-zero control noise run on the launch-per-iteration path: csrc/particle_general.hpp)
    general = "control_type" in g and (str(g["control_type"]) == "velocity" or (not int(g["deterministic"]) and bool(np.any(g["dyn_std"] != 0))))
    general = general or "a_cov" in g  # (full 2 x 2 covariances: launch-per-iteration kernels)
    eligible = "k2" not in name and N % 4 == 0 and D <= 32 and "ctrlpen" not in name and bool(np.all(sp == sp[0])) and not general
    return (T - 1) * calls_per_tick if eligible else 0


class RecordedDraws:
    """The reference's recorded random draws, handed out in call order: `MultiDISCO.draw_source` / `MPF.draw_source` of the mirror API.
    A kind that was not recorded returns None (the library draws); a recorded kind that runs out raises."""

    def __init__(self, eps=None, params=None, ctrl_noise=None, mpf_noise=None):
        self._q = {k: (None if v is None else [np.asarray(a) for a in v]) for k, v in
                   dict(eps=eps, params=params, ctrl_noise=ctrl_noise, mpf_noise=mpf_noise).items()}
        self._i = dict.fromkeys(self._q, 0)

    def _next(self, kind):
        q = self._q[kind]
        if q is None:
            return None
        if self._i[kind] >= len(q):
            raise RuntimeError("recorded draws for %r are exhausted" % kind)
        self._i[kind] += 1
        return q[self._i[kind] - 1]

    def next_eps(self):
        return self._next("eps")

    def next_params(self):
        return self._next("params")

    def next_ctrl_noise(self):
        return self._next("ctrl_noise")

    def next_mpf_noise(self):
        return self._next("mpf_noise")
