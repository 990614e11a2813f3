#!/usr/bin/env python3
"""Writes tests/golden/f64_pin.json: how the canonical float32 arithmetic of this build (oracle/shader_oracle.cpp =
the HIP kernels, bit for bit) compares with the independent float64 restatement of ray_tracer.wgsl
(oracle/independent_f64.py: numpy / libm, true division, no shared header, no BVH) on the Cornell box at 256 x 256.

    python tests/golden/make_f64_pin.py [frames=256]      (build container only: ~6 minutes of numpy)

Recorded: (1) the four debug views that do not count BVH tests, per pixel; (2) frame 0 at the shape of BASELINE
configs[0] (1 spp, 1 bounce) and at 4 bounces: fraction of pixels that agree to 1e-5; (3) the image converged over
`frames` frames (1 spp, 4 bounces) -- per-channel means of the float64 samples, of the float32 oracle's samples with
the SAME per-frame seeds (tight: only the arithmetic differs) and with DISJOINT seeds (frames F .. 2F - 1: a real
two-sample test of equal distributions), against the Monte-Carlo standard error; and the oracle's own progressive
accumulation (wgsl:154-161 in float32) against the float64 mean of its samples.
Rerun (and commit the result) after any change to csrc/rt_transc.h, the oracle or the scene fixture;
tests/test_f64_pin.py checks the committed numbers and re-derives a small live sample on every run."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from oracle import independent_f64 as I  # noqa: E402
from oracle import oracle  # noqa: E402

W = H = 256
BOUNCES = 4


def agree(a, b, rel=1e-5, floor=1e-3):
    """fraction of pixels whose four channels agree to `rel` (relative to max(|b|, floor))"""
    return float((np.abs(a - b) <= rel * np.maximum(np.abs(b), floor)).all(-1).mean())


def oracle_sample(arrays, frame, bounces=BOUNCES, spp=1):
    """the float32 oracle's samples of frame `frame` alone (frames <= 0 stores; |frames| seeds, wgsl:475)"""
    img, _ = oracle.render(rt.make_params(W, H, bounces, spp, skybox=1, frames=-frame), arrays)
    return img


def library_scenes(frames=48, w=160, h=90, spp=2, bounces=5):
    """The scene library's procedural scenes (src/scene/scene.rs: room :445-573, metal :758-863, balls) -- spheres (ray_sphere
    wgsl:223-256), glass (refract, Beer-Lambert, the short-circuit Fresnel draw: wgsl:414-436), mirrors, depth of field -- and
    the texture filter (wgsl:455) on the reference's earthmap.png: the same statistics at a smaller size.  (texture_test's
    FRAMES are not compared: its camera sits exactly on the unit sphere's surface (scene.rs: cam (0, 0, -1), radius 1), so
    whether a ray starts inside or outside is decided by the last bit of |o - c|^2 - r^2 -- in any arithmetic.)"""
    res = {}
    data = os.path.join(ROOT, "tests", "data")
    for name in ("room", "metal", "balls"):
        arrays = rt.SceneArrays.from_scene(rt.Scene.from_name(name, data))
        sc = I.Scene(arrays)
        r = {"width": w, "height": h, "frames": frames, "spp": spp, "bounces": bounces}
        views = {}
        for mode, scale in ((1, 8), (2, 8), (3, 8), (4, 100)):
            p = rt.make_params(w, h, 1, 1, skybox=1, frames=0)
            p.debug_flag, p.debug_scale = mode, scale
            ref, _ = oracle.render(p, arrays)
            got, _hit = I.debug_view(sc, w, h, mode, scale)
            views[f"view{mode}"] = {"max_abs_diff": float(np.abs(got - ref).max()), "hit_mask_mismatches": int(((got[..., 3] != 0) != (ref[..., 3] != 0)).sum())}
        r["debug_views"] = views
        s64 = np.zeros((h, w, 4)); q64 = np.zeros((h, w, 4)); s32 = np.zeros((h, w, 4)); s32o = np.zeros((h, w, 4))
        same = []
        for f in range(frames):
            g = I.render_frame(sc, w, h, bounces, spp, f)
            s64 += g
            q64 += g * g
            a, _ = oracle.render(rt.make_params(w, h, bounces, spp, skybox=1, frames=-f), arrays)
            s32 += a
            same.append(agree(g, a))
            b, _ = oracle.render(rt.make_params(w, h, bounces, spp, skybox=1, frames=-(frames + f)), arrays)
            s32o += b
        m64 = s64 / frames
        var = np.maximum(q64 / frames - m64 * m64, 0.0)
        se = np.sqrt(var.mean((0, 1)) / (w * h * frames))
        r["channel_mean_f64"] = m64.mean((0, 1)).tolist()
        r["standard_error_of_channel_mean"] = se.tolist()
        r["diff_same_seeds_in_standard_errors"] = (np.abs((s64 - s32).mean((0, 1)) / frames) / np.maximum(se, 1e-300)).tolist()
        r["diff_disjoint_seeds_in_standard_errors_of_the_difference"] = (np.abs((s64 - s32o).mean((0, 1)) / frames) / np.maximum(np.sqrt(2.0) * se, 1e-300)).tolist()
        r["per_frame_pixels_within_1e-5_same_seeds"] = {"min": min(same), "mean": float(np.mean(same))}
        res[name] = r
        print(name, json.dumps({k: r[k] for k in r if k != "debug_views"}), flush=True)
    # the texture filter on the reference's own texture (earthmap.png through the loader: fixture)
    tex = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "texture_test_scene.npz")).textures[0]
    rng = np.random.default_rng(5)
    uv = rng.uniform(-2.0, 3.0, (200000, 2)).astype(np.float32)
    got = I.sample_texture(np.asarray(tex), uv[:, 0].astype(np.float64), uv[:, 1].astype(np.float64))
    ref = oracle.sample_texture(tex, uv)
    res["texture_filter_earthmap"] = {"samples": int(uv.shape[0]), "max_abs_diff": float(np.abs(got - ref).max())}
    return res


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
    sc = I.Scene(arrays)
    out = {"scene": "CornellBox-Original (tests/golden/cornell_scene.npz)", "width": W, "height": H, "frames": F, "spp": 1,
           "bounces": BOUNCES}
    views = {}
    for mode, scale in ((1, 8), (2, 8), (3, 8), (4, 100), (4, 400)):
        p = rt.make_params(W, H, 1, 1, skybox=1, frames=0)
        p.debug_flag, p.debug_scale = mode, scale
        ref, _ = oracle.render(p, arrays)
        got, hit = I.debug_view(sc, W, H, mode, scale)
        views[f"view{mode}_scale{scale}"] = {"max_abs_diff": float(np.abs(got - ref).max()),
                                              "hit_mask_mismatches": int((got[..., 3] != ref[..., 3]).sum()),
                                              "hit_fraction": float(hit.mean())}
    out["debug_views"] = views
    f0 = {}
    for nb in (1, 4):
        ref, _ = oracle.render(rt.make_params(W, H, nb, 1, skybox=1, frames=0), arrays)
        got = I.render_frame(sc, W, H, nb, 1, 0)
        f0[f"bounces{nb}"] = {"pixels_within_1e-5": agree(got, ref), "pixels_within_1e-3": agree(got, ref, 1e-3),
                              "mean_diff": (got - ref).mean((0, 1)).tolist()}
    out["frame0"] = f0
    s64 = np.zeros((H, W, 4))
    q64 = np.zeros((H, W, 4))
    s32 = np.zeros((H, W, 4))
    s32_other = np.zeros((H, W, 4))
    acc = np.zeros((H, W, 4), np.float32)
    same = []
    t0 = time.time()
    for f in range(F):
        g = I.render_frame(sc, W, H, BOUNCES, 1, f)
        s64 += g
        q64 += g * g
        r = oracle_sample(arrays, f).astype(np.float64)
        s32 += r
        same.append(agree(g, r))
        s32_other += oracle_sample(arrays, F + f)
        acc, _ = oracle.render(rt.make_params(W, H, BOUNCES, 1, skybox=1, frames=f), arrays, image=acc)
        if f % 16 == 15:
            print(f"frame {f + 1}/{F}  {time.time() - t0:.0f} s", flush=True)
    m64, m32, m32o = s64 / F, s32 / F, s32_other / F
    var = np.maximum(q64 / F - m64 * m64, 0.0)                       # per-pixel variance of one frame's sample
    se_mean = np.sqrt(var.mean((0, 1)) / (W * H * F))               # standard error of a channel's image mean
    se_pixel = np.sqrt(var / F)                                     # ... of one pixel of the converged image
    rgb = slice(0, 4)
    d_same = (m64 - m32).mean((0, 1))
    d_other = (m64 - m32o).mean((0, 1))
    # per pixel, disjoint seeds: the deterministic pixels (the ones that look past the box: the same sky sample every
    # frame) must simply agree.  (No per-pixel z-score for the others: the light is small and the shader samples it by
    # chance only, so a pixel's 256 samples are far too heavy-tailed for their own variance estimate to mean anything.)
    noisy = se_pixel[..., :3] > 1e-9
    det_rel = (np.abs(m64 - m32o) / np.maximum(np.abs(m32o), 1e-3))[..., :3]
    out["converged"] = {
        "channel_mean_f64": m64.mean((0, 1)).tolist(),
        "channel_mean_f32_same_seeds": m32.mean((0, 1)).tolist(),
        "channel_mean_f32_disjoint_seeds": m32o.mean((0, 1)).tolist(),
        "standard_error_of_channel_mean": se_mean.tolist(),
        "diff_same_seeds": d_same.tolist(),
        "diff_same_seeds_in_standard_errors": (np.abs(d_same) / np.maximum(se_mean, 1e-300)).tolist(),
        "diff_disjoint_seeds": d_other.tolist(),
        "diff_disjoint_seeds_in_standard_errors_of_the_difference": (np.abs(d_other) / np.maximum(np.sqrt(2.0) * se_mean, 1e-300)).tolist(),
        "per_frame_pixels_within_1e-5_same_seeds": {"min": min(same), "mean": float(np.mean(same))},
        "converged_pixels_within_1e-5_same_seeds": agree(m64, m32),
        "converged_pixels_within_1e-3_same_seeds": agree(m64, m32, 1e-3),
        "converged_max_relative_diff_same_seeds": float((np.abs(m64 - m32) / np.maximum(np.abs(m32), 1e-3))[..., rgb].max()),
        "disjoint_seeds_noisy_channel_fraction": float(noisy.mean()),
        "disjoint_seeds_deterministic_channels_max_rel_diff": float(det_rel[~noisy].max()) if (~noisy).any() else 0.0,
        "progressive_accumulation_f32_vs_f64_mean_of_its_samples_max_rel": float((np.abs(acc - m32) / np.maximum(np.abs(m32), 1e-3)).max()),
    }
    out["library_scenes"] = library_scenes()
    path = os.path.join(ROOT, "tests", "golden", "f64_pin.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["converged"], indent=1))
    print("wrote", path)


if __name__ == "__main__":
    main()
