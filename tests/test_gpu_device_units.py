"""Device == host for the arithmetic both sides are built from.

The HIP kernels and the CPU oracle compile the same headers (rt_transc.h, rt_texture.h) and rely on
IEEE division / sqrt; whole-image parity would only show a device/host disagreement in them at inputs
the images happen to reach.  Here the device evaluates each building block element-wise
(rt_test_device_units) on >= 10^7 inputs -- the ranges the shader uses plus the edge values it almost
never reaches: rand() outputs that round to 0 and to 1 (-> log(0) = -inf, wgsl:183), subnormals,
+-0, +-inf, NaN, the trig sign-bit shortcut at exact zeros of the reduced argument -- and the result must be
bit-identical to the host compile (oracle.transc).  The shader lines concerned: wgsl:164-206 (RNG,
Box-Muller, disk), :214-221 (pow, smoothstep), :245-251 (acos, atan2), :455 (texture filter).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FN = {"log": 0, "cos": 1, "sin": 2, "exp": 3, "exp2": 4, "log2": 5, "pow": 6, "acos": 7, "atan2": 8, "sqrt": 9,
      "div": 10, "rand": 11, "rng": 12, "trig_signbits": 13, "rand_normal_dist": 14, "rand_convert": 15, "normalize_x": 16, "rcp": 17, "sqrt_dev": 18}
N = 10_000_000
SPECIAL_BITS = np.array([0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007fffff, 0x807fffff, 0x00800000, 0x80800000,
                         0x3f800000, 0xbf800000, 0x3f7fffff, 0x3f800001, 0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000,
                         0x7fc00000, 0xffc00000, 0x7f800001, 0x3f000000, 0x40490fdb, 0x40c90fdb, 0x3fc90fdb, 0x4b800000,
                         0x4f800000, 0x2f800000, 0x33800000, 0x42fe0000, 0xc2fc0000, 0xc3150000, 0x42b17218], np.uint32)


def same_bits(a, b):
    """Bit-identical, except that a NaN equals any NaN: an invalid operation (inf - inf, 0 / 0) produces the
    negative default NaN on x86 and the positive one on gfx950, and NaN operands propagate differently; NaN
    sign and payload are outside the arithmetic contract."""
    a, b = a.view(np.uint32), b.view(np.uint32)
    both_nan = ((a & 0x7fffffff) > 0x7f800000) & ((b & 0x7fffffff) > 0x7f800000)
    bad = np.flatnonzero((a != b) & ~both_nan)
    assert bad.size == 0, f"{bad.size} of {a.size} differ, first at {bad[0]}: device {a[bad[0]]:#010x} host {b[bad[0]]:#010x}"


def with_specials(x):
    return np.concatenate([SPECIAL_BITS.view(np.float32), x.astype(np.float32)])


def rand_values(rng, n):
    """rand() outputs: f32(u32) / 4294967295.0 (wgsl:165), all of [0, 1] including both ends."""
    r = rng.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    return (r.astype(np.float32) / np.float32(4294967295.0)).astype(np.float32)


@pytest.fixture(scope="module")
def rng():
    return np.random.RandomState(20261004)


A_LCG, C_LCG, M_OUT = 747796405, 2891336453, 277803737   # wgsl:195-200


def state_before(output):
    """The RNG state s for which next_random_number(s) returns `output` (the generator is a
    permutation of u32: invert the two xorshifts, the odd multiplications and the LCG step)."""
    w = output ^ (output >> 22)
    x = (w * pow(M_OUT, -1, 2 ** 32)) % 2 ** 32
    k = (x >> 28) + 4                      # the top four bits pass through the xorshift unchanged
    s1, shift = x, k
    while shift < 32:                      # s1 = x ^ (s1 >> k)
        s1 = x ^ (s1 >> k)
        shift += k
    assert ((((s1 >> ((s1 >> 28) + 4)) ^ s1) * M_OUT) % 2 ** 32) == w
    return ((s1 - C_LCG) * pow(A_LCG, -1, 2 ** 32)) % 2 ** 32


def edge_states():
    """States whose next rand() is exactly 0.0 or rounds to 1.0 (r >= 0xffffff80)."""
    return np.array([state_before(0)] + [state_before(r) for r in range(0xffffff80, 0x100000000)], np.uint32)


@pytest.fixture(scope="module")
def tracer(rt):
    """A handle of the TEST library (librt2_mi355x_test.so: the product's sources and flags + the rt_test_* entry points of
    include/rt_test_abi.h, which the product library does not export)."""
    t = rt.RayTracer(device=0, max_width=64, max_height=64, lib=rt.load_test())
    yield t
    t.close()


def test_rand_conversion_over_every_exponent_and_both_ends(tracer, oracle, rng):
    """f32(r) * 2^-32 on the device == f32(r) / 4294967295.0 on the host for raw generator outputs r:
    10^7 random ones, every r that rounds to 2^32 (rand() == 1.0), r = 0 (rand() == 0.0), and the
    neighbours of every power of two."""
    r = np.concatenate([rng.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32),
                        np.arange(0xffffff00, 0x100000000, dtype=np.uint64).astype(np.uint32),
                        np.arange(0, 4096, dtype=np.uint32),
                        np.concatenate([np.array([2 ** k - 1, 2 ** k, 2 ** k + 1], np.uint64) for k in range(1, 32)]).astype(np.uint32)])
    dev = tracer.device_units(FN["rand_convert"], r)
    host = oracle.transc("rand_convert", r)
    same_bits(dev, host)
    assert dev[N:N + 256].max() == np.float32(1.0) and dev[N + 256] == 0.0   # both ends are reached


def test_generator_and_rand_from_states(tracer, oracle, rng):
    s = np.concatenate([rng.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32), np.arange(0, 65536, dtype=np.uint32),
                        np.array([0xffffffff, 0x80000000, 719393, 2073599], np.uint32)])
    same_bits(tracer.device_units(FN["rng"], s), oracle.transc("rng", s))
    same_bits(tracer.device_units(FN["rand"], s), oracle.transc("rand", s))
    e = edge_states()
    dev = tracer.device_units(FN["rand"], e)
    same_bits(dev, oracle.transc("rand", e))
    assert dev[0] == 0.0 and (dev[1:] == 1.0).all()   # rand() is inclusive of both ends


def test_box_muller_from_states(tracer, oracle, rng):
    """rand_normal_dist (wgsl:181-185): theta = 2 pi rand(), rho = sqrt(-2 log(rand())), rho cos(theta).  Includes
    states whose second draw is 0 (log(0) = -inf -> rho = +inf) found by searching the generator."""
    e = edge_states()
    # the second draw of rand_normal_dist is the edge value when the FIRST draw starts one LCG step earlier
    before = np.array([((int(v) - C_LCG) * pow(A_LCG, -1, 2 ** 32)) % 2 ** 32 for v in e], np.uint32)
    s = np.concatenate([rng.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32), e, before])
    dev = tracer.device_units(FN["rand_normal_dist"], s)
    same_bits(dev, oracle.transc("rand_normal_dist", s))
    assert np.isinf(dev[N + e.size]) or np.isnan(dev[N + e.size])   # rho = sqrt(-2 log(0)) = +inf times cos(theta)


@pytest.mark.parametrize("fn", ["log", "log2", "sqrt"])
def test_log_sqrt_on_rand_outputs_and_specials(tracer, oracle, rng, fn):
    x = with_specials(np.concatenate([rand_values(rng, N), np.exp(rng.uniform(-87, 88, 200000)).astype(np.float32)]))
    same_bits(tracer.device_units(FN[fn], x), oracle.transc(fn, x))
    # rand() == 0 -> log(0) = -inf and sqrt(-2 * -inf) = +inf, as the shader's Box-Muller would see it
    zero = np.zeros(1, np.float32)
    assert np.isneginf(tracer.device_units(FN["log"], zero))[0]


@pytest.mark.parametrize("fn", ["cos", "sin", "trig_signbits"])
def test_trig_on_angles_and_specials(tracer, oracle, rng, fn):
    ang = (np.float32(6.2831850051879883) * rand_values(rng, N)).astype(np.float32)              # wgsl:182
    disk = ((rand_values(rng, 1_000_000) * np.float32(2.0)) * np.float32(3.1415926)).astype(np.float32)  # wgsl:203
    # multiples of pi/2 rounded to f32 and their neighbours: where the sign-bit shortcut could disagree
    k = np.arange(0, 9, dtype=np.float64) * (np.pi / 2)
    near = np.concatenate([np.nextafter(k.astype(np.float32), np.float32(d)) for d in (-1e9, 1e9)] + [k.astype(np.float32)])
    x = with_specials(np.concatenate([ang, disk, near, rng.uniform(-100, 100, 500000).astype(np.float32)]))
    same_bits(tracer.device_units(FN[fn], x), oracle.transc(fn, x))


def test_exp_exp2_pow(tracer, oracle, rng):
    x = with_specials(rng.uniform(-110, 110, N).astype(np.float32))
    same_bits(tracer.device_units(FN["exp"], x), oracle.transc("exp", x))
    same_bits(tracer.device_units(FN["exp2"], x), oracle.transc("exp2", x))
    # pow as the shader uses it: smoothstep outputs ^ 0.35, sun cosine ^ 500, (1 - cos)^5 (wgsl:208-221)
    base = with_specials(np.concatenate([rand_values(rng, N // 2), rng.uniform(0, 1.1, N // 2).astype(np.float32)]))
    for e in (0.35, 500.0, 5.0, 0.0, -1.5):
        y = np.full_like(base, e)
        same_bits(tracer.device_units(FN["pow"], base, y), oracle.transc("pow", base, y))


def test_acos_atan2(tracer, oracle, rng):
    x = with_specials(rng.uniform(-1.001, 1.001, N).astype(np.float32))
    same_bits(tracer.device_units(FN["acos"], x), oracle.transc("acos", x))
    a = with_specials(rng.uniform(-1, 1, N).astype(np.float32))
    b = with_specials(rng.uniform(-1, 1, N).astype(np.float32))
    same_bits(tracer.device_units(FN["atan2"], a, b), oracle.transc("atan2", a, b))
    # every pair of special values (signed zeros, infinities, NaN: the quadrant rules)
    sa, sb = np.meshgrid(SPECIAL_BITS.view(np.float32), SPECIAL_BITS.view(np.float32))
    same_bits(tracer.device_units(FN["atan2"], sa.ravel(), sb.ravel()), oracle.transc("atan2", sa.ravel(), sb.ravel()))


def test_division_sqrt_normalize_are_correctly_rounded_on_both_sides(tracer, oracle, rng):
    a = with_specials(np.exp(rng.uniform(-80, 80, N)).astype(np.float32) * rng.choice([-1, 1], N).astype(np.float32))
    b = with_specials(np.exp(rng.uniform(-80, 80, N)).astype(np.float32) * rng.choice([-1, 1], N).astype(np.float32))
    same_bits(tracer.device_units(FN["div"], a, b), oracle.transc("div", a, b))
    sa, sb = np.meshgrid(SPECIAL_BITS.view(np.float32), SPECIAL_BITS.view(np.float32))
    same_bits(tracer.device_units(FN["div"], sa.ravel(), sb.ravel()), oracle.transc("div", sa.ravel(), sb.ravel()))
    # subnormal quotients and divisors
    c = (rng.uniform(1, 2, 1_000_000) * 2.0 ** rng.randint(-149, -100, 1_000_000)).astype(np.float32)
    d = (rng.uniform(1, 2, 1_000_000) * 2.0 ** rng.randint(-30, 30, 1_000_000)).astype(np.float32)
    same_bits(tracer.device_units(FN["div"], c, d), oracle.transc("div", c, d))
    same_bits(tracer.device_units(FN["div"], d, c), oracle.transc("div", d, c))
    u = rng.uniform(-3, 3, N // 4).astype(np.float32)
    v = rng.uniform(-3, 3, N // 4).astype(np.float32)
    same_bits(tracer.device_units(FN["normalize_x"], u, v), oracle.transc("normalize_x", u, v))


def test_rcp_is_the_ieee_division_for_every_float(tracer, oracle, rng):
    """The kernels' reciprocal (v_rcp_f32 + one Newton step in fma arithmetic where x and 1/x are normal and away from
    the denormals, the compiler's IEEE division for the rest of the wave otherwise) is 1.0f / x: the short form is
    compared with the IEEE division ON THE DEVICE for every float it serves, and the whole function with the CPU's
    division on samples across the range, the edge of the range, the specials and waves that mix both paths."""
    checked, bad, first = tracer.sweep(0)
    # 2 signs x (240 binades [2^-120, 2^120) x 2^23 mantissas + the value 2^120)
    assert checked == 2 * (240 * 2 ** 23 + 1) and bad == 0, (checked, bad, hex(first))
    x = with_specials(np.exp(rng.uniform(-88, 88, N)).astype(np.float32) * rng.choice([-1, 1], N).astype(np.float32))
    one = np.ones_like(x)
    same_bits(tracer.device_units(FN["rcp"], x), oracle.transc("div", one, x))
    edge = np.float32(2.0) ** np.float32([-121, -120, -119, 119, 120, 121, -126, -127, 126, 127])
    edge = np.concatenate([np.nextafter(edge, np.float32(0)), edge, np.nextafter(edge, np.float32(np.inf))])
    edge = np.concatenate([edge, -edge])
    mixed = np.where(rng.randint(0, 64, N // 10) == 0, rng.choice(np.concatenate([edge, SPECIAL_BITS.view(np.float32)]), N // 10),
                     rng.uniform(-4, 4, N // 10).astype(np.float32)).astype(np.float32)
    for v in (edge, mixed):
        same_bits(tracer.device_units(FN["rcp"], v), oracle.transc("div", np.ones_like(v), v))


@pytest.mark.parametrize("which,name,at_least", [(2, "sky_gradient_t", 2_100_000_000), (3, "ground_to_sky_t", 2_100_000_000),
                                                   (4, "sun_term", 1_060_000_000)])
def test_sky_shortcuts_equal_the_literal_forms_for_every_float(tracer, which, name, at_least):
    """get_environment_light (wgsl:214-221) as the kernels evaluate it -- smoothstep's clamp and pow's two ends taken as
    branches, so that a wave whose escaping rays look down skips both logarithm / exponential pairs and both divisions
    -- against the literal formulas (which the oracle evaluates), on the device, for EVERY float the shader can hand them:
    dir.y in [-1.5, 1.5] (a unit vector's component, generously), max(0, dot(dir, (0.1, 1, 0.1))) in [0, 1.5]."""
    checked, bad, first = tracer.sweep(which)
    assert checked >= at_least, (name, checked)
    assert bad == 0, (name, bad, hex(first))


def test_sqrt_dev_is_the_ieee_square_root_for_every_float(tracer, oracle, rng):
    """The same for the kernels' square root (v_sqrt_f32 + the neighbour test, without the scaling for denormals)."""
    checked, bad, first = tracer.sweep(1)
    assert checked == 200 * 2 ** 23 + 1 and bad == 0, (checked, bad, hex(first))   # [2^-100, 2^100]
    x = with_specials(np.exp(rng.uniform(-88, 88, N)).astype(np.float32))
    same_bits(tracer.device_units(FN["sqrt_dev"], x), oracle.transc("sqrt", x))
    edge = np.float32(2.0) ** np.float32([-101, -100, -99, 99, 100, 101, -126, -127, 126, 127])
    edge = np.concatenate([np.nextafter(edge, np.float32(0)), edge, np.nextafter(edge, np.float32(np.inf)), -edge])
    mixed = np.where(rng.randint(0, 64, N // 10) == 0, rng.choice(np.concatenate([edge, SPECIAL_BITS.view(np.float32)]), N // 10),
                     rng.uniform(0, 4, N // 10).astype(np.float32)).astype(np.float32)
    for v in (edge, mixed):
        same_bits(tracer.device_units(FN["sqrt_dev"], v), oracle.transc("sqrt", v))


def test_texture_filter(tracer, oracle, rng):
    """textureSampleLevel as rt_texture.h defines it: sRGB table, bilinear weights, repeat addressing, on
    and far beyond [0, 1), plus non-finite coordinates."""
    for shape in ((16, 8), (5, 3), (1, 1), (64, 64)):
        tex = rng.randint(0, 256, shape + (4,), dtype=np.uint8)
        uv = np.concatenate([rng.uniform(-3, 4, (400000, 2)), rng.uniform(-1e6, 1e6, (1000, 2)),
                             np.array([[0, 0], [1, 1], [0.5, 0.5], [-0.0, 1.0], [np.inf, 0.2], [0.3, -np.inf], [np.nan, np.nan]])]).astype(np.float32)
        same_bits(tracer.device_sample_texture(tex, uv).ravel(), oracle.sample_texture(tex, uv).ravel())
