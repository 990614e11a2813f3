"""Generates the committed fixtures under tests/golden/.

Run in the build container (needs /root/reference/assets for the OBJ/MTL and
texture DATA files only; no reference code is imported or executed):

    python tests/golden/make_golden.py

PARITY UNPINNED: the reference ships no golden vectors and cannot be run
(SURVEY.md 8c), so these fixtures are produced by this repo's own host
pipeline (scene arrays) and CPU oracle (images, transcripts).  They pin the
oracle against regressions and let the GPU box, where /root/reference does not
exist, load the scenes.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import ray_tracer_2_amd as rt  # noqa: E402
from oracle import oracle  # noqa: E402

ASSETS = "/root/reference/assets"


def main():
    # (1) Cornell scene arrays (8 meshes / 32 triangles / 32 nodes)
    sc = rt.Scene.from_name("cornell_box", ASSETS)
    arr = rt.SceneArrays.from_scene(sc)
    assert arr.meshes.shape[0] == 8 and arr.triangles.shape[0] == 32 and arr.nodes.shape[0] == 32
    arr.save(os.path.join(HERE, "cornell_scene.npz"))

    out = {}
    # (3) RNG known-answer vectors
    for seed in (0, 1, 2073599, 719393):
        out[f"rng_{seed}"] = oracle.rng_sequence(seed, 16)
        out[f"rand_{seed}"] = oracle.rand_sequence(seed, 16)
    # (2) debug views at 64x36
    for mode in range(1, 8):
        for scale in (8, 100):
            p = rt.make_params(64, 36, 4, 1, skybox=1, frames=0, debug_flag=mode, debug_scale=scale)
            img, _ = oracle.render(p, arr)
            out[f"debug_{mode}_{scale}"] = img
    # (5) full frames at config-1 shape
    for sky in (1, 0):
        p = rt.make_params(256, 256, 1, 1, skybox=sky, frames=0)
        img, st = oracle.render(p, arr)
        out[f"frame_256_sky{sky}"] = img
        out[f"frame_256_sky{sky}_segments"] = np.array([st.segments], dtype=np.uint64)
    # (6) accumulation frames 0,1,2 at 64x36, config-2 sampling
    img = np.zeros((36, 64, 4), np.float32)
    for f in range(3):
        p = rt.make_params(64, 36, 4, 8, skybox=1, frames=f)
        img, _ = oracle.render(p, arr, image=img)
        out[f"accum_{f}"] = img.copy()
    # (4) per-pixel path transcripts at config-1 shape
    rng = np.random.RandomState(7)
    pix = np.stack([rng.randint(0, 256, 32), rng.randint(0, 256, 32)], axis=1).astype(np.uint32)
    p = rt.make_params(256, 256, 1, 1, skybox=1, frames=0)
    recs, vals = [], []
    for x, y in pix:
        rgba, rec = oracle.trace_pixel(p, arr, int(x), int(y))
        pad = np.zeros(4, dtype=oracle.TRANSCRIPT_DTYPE)
        pad[:len(rec)] = rec
        recs.append(pad)
        vals.append(rgba)
    out["transcript_pixels"] = pix
    out["transcript_records"] = np.stack(recs)
    out["transcript_rgba"] = np.stack(vals)
    np.savez_compressed(os.path.join(HERE, "cornell_golden.npz"), **out)
    print("wrote", len(out), "arrays")
    library_scenes()


def library_scenes():
    """(7) The two library scenes that run on the reference's own data files and nothing else: `texture_test`
    (scene.rs:280-309: a unit sphere textured with assets/earthmap.png -- the real PNG -> horizontal flip -> sRGB ->
    spherical-uv path of asset.rs:60-80 and wgsl:248-252,454-455) and `obj_test` (scene.rs:310-364: dragon.obj
    without its .mtl, an emissive quad, three spheres).  Scene arrays (with the decoded, flipped texture) + small
    oracle images: the path-traced frame after two accumulated frames and the debug views 1 (normal) and 3 (uv)."""
    out = {}
    for name in ("texture_test", "obj_test"):
        arr = rt.SceneArrays.from_scene(rt.Scene.from_name(name, ASSETS))
        arr.save(os.path.join(HERE, f"{name}_scene.npz"))
        img = np.zeros((54, 96, 4), np.float32)
        for f in range(2):
            img, st = oracle.render(rt.make_params(96, 54, 3, 4, skybox=1, frames=f), arr, image=img)
        out[f"{name}_frame"] = img
        out[f"{name}_segments_frame1"] = np.array([st.segments], np.uint64)
        for dbg in (1, 2, 3):
            out[f"{name}_debug_{dbg}"], _ = oracle.render(rt.make_params(96, 54, 3, 1, debug_flag=dbg, debug_scale=4), arr)
    np.savez_compressed(os.path.join(HERE, "library_scenes_golden.npz"), **out)
    print("wrote the texture_test / obj_test fixtures")


if __name__ == "__main__":
    if "--library-scenes" in sys.argv:   # only (7): leaves the Cornell fixtures alone
        library_scenes()
    else:
        main()
