"""Accuracy of the canonical f32 transcendentals (rt_transc.h) against
double-precision numpy, on the ranges the shader uses, plus special values."""
import numpy as np


def ulp_err(got, ref):
    got = got.astype(np.float64)
    with np.errstate(all="ignore"):
        e = np.floor(np.log2(np.maximum(np.abs(ref), 2.0 ** -126)))
    ulp = 2.0 ** (e - 23)
    return np.abs(got - ref) / ulp


RNG = np.random.RandomState(1234)


def rand_grid(n):
    r = RNG.randint(1, 2 ** 32, size=n, dtype=np.uint64)
    return (r.astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)


def test_log_on_rand_outputs(oracle):
    x = rand_grid(400000)
    assert ulp_err(oracle.transc("log", x), np.log(x.astype(np.float64))).max() <= 1.0


def test_cos_sin_on_angles(oracle):
    x = (np.float32(6.2831850051879883) * rand_grid(400000)).astype(np.float32)
    assert ulp_err(oracle.transc("cos", x), np.cos(x.astype(np.float64))).max() <= 2.0
    assert ulp_err(oracle.transc("sin", x), np.sin(x.astype(np.float64))).max() <= 2.0


def test_exp_exp2_log2(oracle):
    x = RNG.uniform(-80, 80, 200000).astype(np.float32)
    assert ulp_err(oracle.transc("exp", x), np.exp(x.astype(np.float64))).max() <= 1.5
    x = RNG.uniform(-120, 120, 200000).astype(np.float32)
    assert ulp_err(oracle.transc("exp2", x), np.exp2(x.astype(np.float64))).max() <= 1.5
    x = np.exp(RNG.uniform(-60, 60, 200000)).astype(np.float32)
    assert ulp_err(oracle.transc("log2", x), np.log2(x.astype(np.float64))).max() <= 1.0


def test_pow_is_exp2_of_y_log2(oracle):
    """WGSL: pow(x, y) = exp2(y * log2(x)); relative error grows with |y log2 x|."""
    x = rand_grid(100000)
    for y in (0.35, 5.0, 500.0):
        got = oracle.transc("pow", x, np.full_like(x, y)).astype(np.float64)
        ref = np.power(x.astype(np.float64), y)
        ok = ref > 1e-30
        rel = np.abs(got[ok] - ref[ok]) / ref[ok]
        assert rel.max() < 2e-7 * (1 + 1.5 * np.abs(y * np.log2(x[ok].astype(np.float64))).max())
    assert oracle.transc("pow", np.float32([0, 0, 1]), np.float32([0.35, 5, 500])).tolist() == [0, 0, 1]


def test_acos_atan2(oracle):
    x = RNG.uniform(-1, 1, 200000).astype(np.float32)
    assert ulp_err(oracle.transc("acos", x), np.arccos(x.astype(np.float64))).max() <= 2.5
    y = RNG.uniform(-1, 1, 200000).astype(np.float32)
    got, ref = oracle.transc("atan2", y, x), np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.abs(got - ref).max() < 1e-6


def test_special_values(oracle):
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    assert oracle.transc("log", np.float32([0.0]))[0] == -inf           # log(rand() == 0)
    assert np.isnan(oracle.transc("log", np.float32([-1.0]))[0])
    assert oracle.transc("log", np.float32([1.0]))[0] == 0
    assert oracle.transc("cos", np.float32([0.0]))[0] == 1 and oracle.transc("sin", np.float32([0.0]))[0] == 0
    assert np.isnan(oracle.transc("cos", np.float32([inf]))[0])
    assert oracle.transc("exp", np.float32([-inf, 0, 100]))[:2].tolist() == [0, 1]
    assert oracle.transc("exp", np.float32([100]))[0] == inf
    assert oracle.transc("exp2", np.float32([-150, -149, 127]))[0] >= 0
    assert np.isnan(oracle.transc("acos", np.float32([1.5]))[0])
    assert np.isnan(oracle.transc("sqrt", np.float32([-1]))[0]) and np.isnan(oracle.transc("log", np.float32([nan]))[0])


def test_normalize_stays_within_the_wgsl_bound(oracle):
    """normalize(v) = v * (1 / sqrt(dot(v, v))) (one correctly rounded reciprocal of the correctly rounded
    length, then a multiplication).  WGSL: normalize's accuracy is "inherited from e / length(e)", length's from
    sqrt(dot(e, e)), and a division is good to 2.5 ULP -- so the bound is 2.5 ULP against e divided EXACTLY by the
    binary32 length (the dot product in the shader's order, the correctly rounded square root).  Measured: 1.5.
    Against the exact normalised vector the whole chain stays within 3 ULP."""
    n = 400000
    x = np.exp(RNG.uniform(-18, 18, n)).astype(np.float32) * RNG.choice([-1, 1], n).astype(np.float32)
    y = np.exp(RNG.uniform(-18, 18, n)).astype(np.float32) * RNG.choice([-1, 1], n).astype(np.float32)
    z = (x * y).astype(np.float32)
    length32 = np.sqrt(((x * x).astype(np.float32) + (y * y).astype(np.float32)).astype(np.float32) + (z * z).astype(np.float32),
                       dtype=np.float32)
    ok = np.isfinite(length32) & (length32 > 1e-30)
    got = oracle.transc("normalize_x", x, y)
    assert ulp_err(got[ok], x[ok].astype(np.float64) / length32[ok].astype(np.float64)).max() <= 2.5
    xd, yd, zd = (v.astype(np.float64) for v in (x, y, z))
    assert ulp_err(got[ok], (xd / np.sqrt(xd * xd + yd * yd + zd * zd))[ok]).max() <= 3.0


def test_log_on_every_rand_output(oracle):
    """Box-Muller takes log(rand()) (wgsl:183) and rand() = f32(r) * 2^-32 takes every float of [2^-32, 1] (and 0): the
    bound of test_log_on_rand_outputs on ALL 2.7e8 of them, not a sample (the polynomial was refitted in round 3)."""
    worst = 0.0
    for e in range(-32, 0):   # one binade at a time: 2^23 floats each
        x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32((e + 127) << 23)).view(np.float32)
        worst = max(worst, float(ulp_err(oracle.transc("log", x), np.log(x.astype(np.float64))).max()))
    assert worst <= 1.0, worst
    assert oracle.transc("log", np.float32([1.0]))[0] == 0.0


def test_sun_term_pow_500(oracle):
    """get_environment_light's sun (wgsl:218): pow(max(0, dot(dir, (0.1, 1, 0.1))), 500).  (a) For every float in
    [0.8, 1.02] -- the only inputs with a non-zero result: dot <= |dir| * |(0.1, 1, 0.1)| = 1.00995 -- the relative error
    against double precision; (b) below 0.8 the result is exactly +0 (exp2_ returns +0 under -150: what the kernels'
    early-out relies on; every float is compared on the device, here every 64th)."""
    lo, hi = np.float32(0.8).view(np.uint32), np.float32(1.02).view(np.uint32)
    x = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)
    got = oracle.transc("pow", x, np.full_like(x, 500.0)).astype(np.float64)
    ref = np.power(x.astype(np.float64), 500.0)
    ok = ref > 1e-35   # (below: denormal results)
    rel = np.abs(got[ok] - ref[ok]) / ref[ok]
    # y * log2(x) reaches 161 in magnitude: one ulp of log2 is worth 161 * 2^-24 * ln 2 in the result
    assert rel.max() < 500 * 2.0 ** -23, rel.max()
    below = np.arange(0, lo, 64, dtype=np.uint32).view(np.float32)
    assert not oracle.transc("pow", below, np.full_like(below, 500.0)).view(np.uint32).any()
