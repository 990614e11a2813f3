"""A pin that does not share the kernels' headers: the canonical float32 arithmetic (oracle/shader_oracle.cpp, which the
HIP kernels equal bit for bit) against an independent float64 restatement of shaders/ray_tracer.wgsl
(oracle/independent_f64.py: numpy / libm, true division, the shader's literal constants, no BVH).

north_star asks for pixels "within 1e-5 of the reference render"; the reference cannot be built or run here (SURVEY.md
8c), and across arithmetic implementations a path tracer only agrees statistically (one flipped roulette
decision changes a sample by O(1)).  What can be pinned, and is: the deterministic outputs (the debug views of the
primary hit, wgsl:502-573) agree per pixel to 1e-5; a frame's samples agree to 1e-5 on all but the handful of pixels
where a decision flipped; and the converged image has the same per-channel means within the Monte-Carlo standard error,
with the same seeds (tight) and with disjoint seeds (a two-sample test).  tests/golden/f64_pin.json holds the 256-frame
statistics at 256 x 256 (tests/golden/make_f64_pin.py, minutes of numpy: build container only); a small live sample is
re-derived on every run so that a change to csrc/rt_transc.h cannot slip through unnoticed."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def pin():
    return json.load(open(os.path.join(GOLDEN, "f64_pin.json")))


def test_committed_statistics_debug_views_agree_per_pixel(pin):
    for name, v in pin["debug_views"].items():
        assert v["hit_mask_mismatches"] == 0, name          # the same pixels hit the scene
        assert v["max_abs_diff"] <= 1e-5, (name, v)          # north_star's tolerance (values are O(1))


def test_committed_statistics_frames_and_converged_image(pin):
    assert pin["frames"] >= 256 and (pin["width"], pin["height"]) == (256, 256)
    for k, v in pin["frame0"].items():
        assert v["pixels_within_1e-5"] >= 0.995, (k, v)      # all but the pixels where a decision flipped
        assert v["pixels_within_1e-3"] >= 0.999, (k, v)
    c = pin["converged"]
    assert c["per_frame_pixels_within_1e-5_same_seeds"]["min"] >= 0.995
    # the same seeds: only the arithmetic differs -- a small fraction of one standard error
    assert max(c["diff_same_seeds_in_standard_errors"]) <= 0.5
    # disjoint seeds: two independent estimates of the same image
    assert max(c["diff_disjoint_seeds_in_standard_errors_of_the_difference"]) <= 3.5
    assert c["disjoint_seeds_deterministic_channels_max_rel_diff"] <= 1e-5
    # wgsl:154-161 in float32 over 256 frames drifts by a few ulps from the exact mean of the same samples
    assert c["progressive_accumulation_f32_vs_f64_mean_of_its_samples_max_rel"] <= 1e-5


def test_live_sample_against_the_float64_restatement(rt, oracle, cornell):
    from oracle import independent_f64 as I
    W, H = 96, 54
    sc = I.Scene(cornell)
    for mode, scale in ((1, 8), (2, 8), (3, 8), (4, 100), (4, 400)):
        p = rt.make_params(W, H, 1, 1, skybox=1, frames=0)
        p.debug_flag, p.debug_scale = mode, scale
        ref, _ = oracle.render(p, cornell)
        got, _hit = I.debug_view(sc, W, H, mode, scale)
        assert np.array_equal(got[..., 3] != 0, ref[..., 3] != 0), mode
        assert np.abs(got - ref).max() <= 1e-5, (mode, scale)
    s64 = np.zeros((H, W, 4))
    s32 = np.zeros((H, W, 4))
    for f in range(6):
        g = I.render_frame(sc, W, H, 4, 2, f)
        r, _ = oracle.render(rt.make_params(W, H, 4, 2, skybox=1, frames=-f), cornell)
        ok = (np.abs(g - r) <= 1e-5 * np.maximum(np.abs(r), 1e-3)).all(-1)
        assert ok.mean() >= 0.99, (f, ok.mean())
        s64 += g
        s32 += r
    # same seeds: the image means differ by far less than the image mean's own Monte-Carlo error (~1e-2 here)
    assert np.abs((s64 - s32).mean((0, 1)) / 6).max() <= 2e-3


def test_committed_statistics_library_scenes(pin):
    """Spheres, glass, mirrors, depth of field (the scene library's room / metal / balls) and the texture filter on the
    reference's earthmap.png: the same comparison.  The per-pixel tolerances are wider than on the Cornell box because
    ray_sphere's quadratic (wgsl:229-240) cancels in binary32 for big or distant spheres -- |o - c|^2 - r^2 loses up to
    five digits on the radius-10 ground sphere -- in ANY binary32 implementation, the reference's included."""
    lib = pin["library_scenes"]
    for name in ("room", "metal", "balls"):
        r = lib[name]
        for view, v in r["debug_views"].items():
            assert v["hit_mask_mismatches"] == 0 and v["max_abs_diff"] <= 2e-4, (name, view, v)
        assert r["per_frame_pixels_within_1e-5_same_seeds"]["min"] >= 0.97, name     # (metal: chains of mirror bounces)
        assert max(r["diff_same_seeds_in_standard_errors"]) <= 2.0, name
        assert max(r["diff_disjoint_seeds_in_standard_errors_of_the_difference"]) <= 3.5, name
    assert lib["texture_filter_earthmap"]["max_abs_diff"] <= 2e-4     # (u * width - 0.5 in binary32 at |u| up to 3)


def test_live_sample_with_spheres_and_glass(rt, oracle):
    from conftest import ROOT
    from oracle import independent_f64 as I
    W, H = 96, 54
    arrays = rt.SceneArrays.from_scene(rt.Scene.from_name("room", os.path.join(ROOT, "tests", "data")))
    assert arrays.spheres.shape[0] > 0 and any(int(sp["material"]["flag"]) == 1 for sp in arrays.spheres)   # a glass sphere
    sc = I.Scene(arrays)
    for mode in (1, 2, 3):
        p = rt.make_params(W, H, 1, 1, skybox=1, frames=0)
        p.debug_flag, p.debug_scale = mode, 8
        ref, _ = oracle.render(p, arrays)
        got, _hit = I.debug_view(sc, W, H, mode, 8)
        assert np.array_equal(got[..., 3] != 0, ref[..., 3] != 0) and np.abs(got - ref).max() <= 2e-4, mode
    for f in range(3):
        g = I.render_frame(sc, W, H, 5, 2, f)
        r, _ = oracle.render(rt.make_params(W, H, 5, 2, skybox=1, frames=-f), arrays)
        assert (np.abs(g - r) <= 1e-5 * np.maximum(np.abs(r), 1e-3)).all(-1).mean() >= 0.99, f
