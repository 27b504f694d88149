"""Coefficients of pendulum_trig (dust_amd/csrc/handoff.hpp): sin / cos on [-pi/2 - d, pi/2 + d] as odd / even polynomials in r
(reduction by pi, sign from the parity of the multiple), fitted by Remez-style reweighted least squares in float64, rounded to
float32, and checked in emulated float32 arithmetic (every operation rounded once, fma = one rounding) against float64 sin / cos -
next to the [-pi/4, pi/4] pair of round 1-3 (common.hpp poly_sin / poly_cos).   python tools/trig_fit.py"""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def fit(kind, ncoef, hi, iters=60):
    """minimax fit of sin(r) = r + r s P(s) / cos(r) = 1 + s Q(s), s = r^2, absolute error on [0, hi]"""
    r = np.cos(np.linspace(0, np.pi, 4001)) * 0.5 * hi + 0.5 * hi  # Chebyshev-spaced points of [0, hi]
    r = r[r > 1e-9]
    s = r * r
    if kind == "sin":
        target, basis_scale = np.sin(r) - r, r * s
    else:
        target, basis_scale = np.cos(r) - 1.0, s
    A = np.stack([basis_scale * s ** k for k in range(ncoef)], axis=1)
    w = np.ones_like(r)
    best = None
    for _ in range(iters):
        c, *_ = np.linalg.lstsq(A * w[:, None], target * w, rcond=None)
        err = A @ c - target
        m = np.abs(err).max()
        if best is None or m < best[0]:
            best = (m, c)
        w = w * (1.0 + 0.5 * (np.abs(err) / m) ** 2)  # Lawson-style reweighting towards the equi-oscillating solution
        w /= w.max()
    return best


def eval_new(th, S, C):
    """emulated float32: reduction by pi with the rounding magic, odd / even polynomials, sign by parity"""
    th = th.astype(f32)
    magic = f32(12582912.0)
    t = fma(th, f32(0.318309886183790671538), magic)
    kf = (t - magic).astype(f32)
    sign = np.where((t.view(np.uint32) & 1) == 1, f32(-1), f32(1))
    r = fma(kf, f32(-3.14159202575683593750), th)  # 2 x the pi/2 constants of trig_reduce (exact doublings)
    r = fma(kf, f32(-6.27832946e-07), r)  # (the third term, 1.08e-14 k, changes nothing below |theta| = 5e4)
    s = (r * r).astype(f32)
    p = f32(S[-1]) * np.ones_like(s)
    for c in S[-2::-1]:
        p = fma(p, s, f32(c))
    tt = (r * s).astype(f32)
    sn = fma(p, tt, r)
    q = f32(C[-1]) * np.ones_like(s)
    for c in C[-2::-1]:
        q = fma(q, s, f32(c))
    cs = fma(q, s, f32(1.0))
    return (sn * sign).astype(f32), (cs * sign).astype(f32)


def eval_old(th):
    th = th.astype(f32)
    k = np.rint((th * f32(0.636619747)).astype(f32)).astype(f32)
    q = k.astype(np.int64)
    r = fma(k, f32(-1.57079601e+00), th)
    r = fma(k, f32(-3.13916473e-07), r)
    r = fma(k, f32(-5.39030253e-15), r)
    s = (r * r).astype(f32)
    p = f32(2.86567956e-6) * np.ones_like(s)
    for c in (-1.98559923e-4, 8.33338592e-3, -1.66666672e-1):
        p = fma(p, s, f32(c))
    ps = fma(p, (r * s).astype(f32), r)
    p = f32(2.44677067e-5) * np.ones_like(s)
    for c in (-1.38877297e-3, 4.16666567e-2, -5.00000000e-1):
        p = fma(p, s, f32(c))
    pc = fma(p, s, f32(1.0))
    sn = np.where(q & 1, pc, ps)
    cs = np.where(q & 1, ps, pc)
    sn = np.where(q & 2, -sn, sn)
    cs = np.where((q + 1) & 2, -cs, cs)
    return sn.astype(f32), cs.astype(f32)


if __name__ == "__main__":
    hi = np.pi / 2 + 0.02
    es, S = fit("sin", 4, hi)
    ec, C = fit("cos", 5, hi)
    S32, C32 = [f32(c) for c in S], [f32(c) for c in C]
    print("sin fit error (float64 coefficients) %.3e   cos %.3e" % (es, ec))
    print("S =", ", ".join("%.9ef" % c for c in S32))
    print("C =", ", ".join("%.9ef" % c for c in C32))
    rng = np.random.default_rng(0)
    for span in (4.0, 40.0, 1000.0, 5.0e4):
        th = (rng.uniform(-span, span, 2_000_000)).astype(f32)
        ref_s, ref_c = np.sin(th.astype(np.float64)), np.cos(th.astype(np.float64))
        ns, nc = eval_new(th, S32, C32)
        os_, oc = eval_old(th)
        ulp = lambda v, ref: np.abs(v.astype(np.float64) - ref) / np.spacing(np.abs(ref).astype(f32)).astype(np.float64)
        print("|theta| <= %-8g new: max abs err sin %.2e cos %.2e (ulp %.1f / %.1f) | old: sin %.2e cos %.2e (ulp %.1f / %.1f)" % (
            span, np.abs(ns - ref_s).max(), np.abs(nc - ref_c).max(), ulp(ns, ref_s).max(), ulp(nc, ref_c).max(),
            np.abs(os_ - ref_s).max(), np.abs(oc - ref_c).max(), ulp(os_, ref_s).max(), ulp(oc, ref_c).max()))
