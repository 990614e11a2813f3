"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same inputs.  The bar is BIT-EXACT images (stricter than north_star's 1e-5
relative): both sides evaluate the same IEEE binary32 operations, so any
difference is a bug.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits

pytestmark = pytest.mark.gpu


def assert_bit_equal(gpu, ref, what):
    g, r = bits(gpu), bits(ref)
    if not np.array_equal(g, r):
        bad = np.argwhere(g != r)
        rel = np.abs(gpu.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1e-30)
        raise AssertionError(f"{what}: {len(bad)} of {g.size} values differ; first at {bad[0]}: "
                             f"gpu={gpu[tuple(bad[0])]!r} ref={ref[tuple(bad[0])]!r}; "
                             f"max rel err {np.nanmax(rel):.3e}")


@pytest.fixture(scope="module", params=[(0, 1), (1, 1), (0, 0), (1, 0)],
                ids=["persistent-lds", "tiles-lds", "persistent-global", "tiles-global"])
def loaded(request, tracer, cornell):
    """Every kernel variant x scene placement must produce the same bits."""
    tracer.load_scene(cornell)
    tracer.set_option("kernel_variant", request.param[0])
    tracer.set_option("lds_scene", request.param[1])
    yield tracer
    tracer.set_option("kernel_variant", -1)
    tracer.set_option("lds_scene", 1)


@pytest.mark.parametrize("mode", range(1, 8))
@pytest.mark.parametrize("scale", [8, 100])
def test_debug_views(rt, oracle, loaded, cornell, mode, scale):
    p = rt.make_params(64, 36, 4, 1, debug_flag=mode, debug_scale=scale)
    loaded.render(p)
    gpu = loaded.read_image(64, 36)
    ref, _ = oracle.render(p, cornell)
    assert_bit_equal(gpu, ref, f"debug view {mode}/{scale}")
    gold = np.load(os.path.join(GOLDEN, "cornell_golden.npz"))[f"debug_{mode}_{scale}"]
    assert_bit_equal(gpu, gold, f"debug view {mode}/{scale} vs golden")


@pytest.mark.parametrize("sky", [1, 0])
def test_config1_frame(rt, oracle, loaded, cornell, sky):
    """BASELINE config 1 shape: 256x256, 1 spp, 1 bounce, frames = 0."""
    p = rt.make_params(256, 256, 1, 1, skybox=sky, frames=0)
    loaded.reset_timing()
    loaded.render(p)
    gpu = loaded.read_image(256, 256)
    ref, st = oracle.render(p, cornell)
    assert_bit_equal(gpu, ref, f"config-1 frame sky={sky}")
    s = loaded.stats()
    assert s.segments == st.segments and s.paths == 256 * 256 and s.launches == 1
    gold = np.load(os.path.join(GOLDEN, "cornell_golden.npz"))[f"frame_256_sky{sky}"]
    assert_bit_equal(gpu, gold, "config-1 frame vs golden")


def test_accumulation(rt, oracle, loaded, cornell):
    """frames = 0, 1, 2: plain store, then blends with weight 1/(frames+1) (wgsl:154-161)."""
    ref = np.zeros((36, 64, 4), np.float32)
    gold = np.load(os.path.join(GOLDEN, "cornell_golden.npz"))
    for f in range(3):
        p = rt.make_params(64, 36, 4, 8, frames=f)
        loaded.render(p)
        ref, _ = oracle.render(p, cornell, image=ref)
        gpu = loaded.read_image(64, 36)
        assert_bit_equal(gpu, ref, f"accumulated frame {f}")
        assert_bit_equal(gpu, gold[f"accum_{f}"], f"accumulated frame {f} vs golden")


@pytest.mark.parametrize("shape", [(480, 270), (333, 77), (8, 8), (1, 1), (9, 17)])
def test_config2_sampling_small(rt, oracle, loaded, cornell, shape):
    """Config-2 sampling (8 spp, 4 bounces) at sizes the oracle finishes in seconds,
    including ragged sizes that leave partial 8x8 tiles."""
    w, h = shape
    p = rt.make_params(w, h, 4, 8, frames=0)
    loaded.reset_timing()
    loaded.render(p)
    gpu = loaded.read_image(w, h)
    ref, st = oracle.render(p, cornell)
    assert_bit_equal(gpu, ref, f"{w}x{h} 8spp 4b")
    assert loaded.stats().segments == st.segments


def test_counters_match_oracle(rt, oracle, loaded, cornell):
    p = rt.make_params(128, 72, 4, 2, frames=3)
    loaded.set_counters(True)
    try:
        loaded.write_image(np.zeros((72, 128, 4), np.float32))
        loaded.reset_timing()
        loaded.render(p)
        s = loaded.stats()
    finally:
        loaded.set_counters(False)
    _, st = oracle.render(p, cornell)
    assert (s.segments, s.node_tests, s.triangle_tests) == (st.segments, st.node_tests, st.triangle_tests)


def test_export_of_a_rendered_frame(rt, oracle, loaded, cornell):
    """SURVEY 8(f)-3 on the device's output: rt_read_image + rt_export_rgba8 (app.rs:341-460) against the
    oracle's frame through the oracle's restatement of the export loop."""
    w, h = 96, 54
    p = rt.make_params(w, h, 4, 4, skybox=1, frames=0)
    loaded.render(p)
    gpu = loaded.read_image(w, h)
    out = np.zeros((h, w, 4), np.uint8)
    assert rt.load().rt_export_rgba8(gpu.ctypes.data, w, h, out.ctypes.data) == 0
    ref, _ = oracle.render(p, cornell)
    assert np.array_equal(out, oracle.export_rgba8(ref))
