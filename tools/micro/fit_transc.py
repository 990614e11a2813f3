#!/usr/bin/env python3
"""Derivation record of the polynomial kernels in csrc/rt_transc.h (round 3: cheaper canonical cos / sin / log).

Near-minimax fits in float64 (weighted least squares on dense Chebyshev points, a few Lawson re-weighting steps),
coefficients rounded to float32, then the kernels are evaluated in emulated float32 arithmetic over every reduced
argument that matters and compared with float64 libm.  The authoritative error figures come from the compiled
header itself (tests/test_transc.py runs the oracle's build of rt_transc.h over dense grids).

    python tools/micro/fit_transc.py
"""
import numpy as np

f32 = np.float32


def lawson_fit(x, target, basis, weight, iters=60):
    """min max |weight * (basis @ c - target)| by iteratively re-weighted least squares."""
    w = np.ones_like(x)
    c = None
    for _ in range(iters):
        A = basis * (weight * np.sqrt(w))[:, None]
        b = target * weight * np.sqrt(w)
        c, *_ = np.linalg.lstsq(A, b, rcond=None)
        err = np.abs(weight * (basis @ c - target))
        w = w * (err / err.max() + 1e-3)
        w /= w.sum()
    return c, np.abs(weight * (basis @ c - target)).max()


def cheb_points(a, b, n):
    k = np.arange(n)
    return 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (k + 0.5) / n)


def sincos(deg_s=4, deg_c=4):
    rmax = np.pi / 4 * 1.0005
    r = np.sort(cheb_points(1e-4, rmax, 4000))
    z = r * r
    # sin(r) = r + r z (S1 + S2 z + ...): fit g(z) = (sin(r)/r - 1)/z, relative error of sin = z * err(g)
    g = (np.sin(r) / r - 1.0) / z
    B = np.stack([z ** k for k in range(deg_s)], 1)
    S, es = lawson_fit(z, g, B, z)
    # cos(r) = 1 + z (C0 + C1 z + ...): fit h(z) = (cos(r) - 1)/z, absolute error of cos = z * err(h)
    h = (np.cos(r) - 1.0) / z
    B = np.stack([z ** k for k in range(deg_c)], 1)
    C, ec = lawson_fit(z, h, B, z)
    print("sin coefficients S1..:", [float(f32(v)) for v in S], "fit rel err %.3g (ulp of ~1: %.3g)" % (es, es / 2 ** -24))
    print("cos coefficients C0..:", [float(f32(v)) for v in C], "fit abs err %.3g (ulp in [0.5,1): %.3g)" % (ec, ec / 2 ** -24))
    return S.astype(f32), C.astype(f32)


def logk(deg=8):
    # log(1 + f) = f - f^2/2 + f^3 P(f), f in [sqrt(1/2) - 1, sqrt(2) - 1]
    a, b = np.sqrt(0.5) - 1, np.sqrt(2.0) - 1
    f = np.sort(cheb_points(a * 1.0005, b * 1.0005, 6000))
    f = f[np.abs(f) > 1e-3]
    t = (np.log1p(f) - f + 0.5 * f * f) / f ** 3
    B = np.stack([f ** k for k in range(deg + 1)], 1)
    # relative error of log(1+f) (k = 0 is the worst case: nothing else to hide behind)
    P, e = lawson_fit(f, t, B, np.abs(f ** 3 / np.log1p(f)))
    print("log coefficients P0..:", [float(f32(v)) for v in P], "fit rel err %.3g (ulp: %.3g)" % (e, e / 2 ** -24))
    return P.astype(f32)


if __name__ == "__main__":
    for ds, dc in ((4, 4), (4, 5), (5, 5)):
        print("degrees", ds, dc)
        sincos(ds, dc)
    for d in (6, 7, 8):
        print("degree", d)
        logk(d)
