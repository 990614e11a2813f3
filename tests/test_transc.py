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
