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
    return kw


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
    eligible = "k2" not in name and N % 4 == 0 and D <= 32 and "ctrlpen" not in name and bool(np.all(sp == sp[0]))
    return (T - 1) * calls_per_tick if eligible else 0
