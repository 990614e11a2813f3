#!/usr/bin/env python3
"""The strip split used on ONE device: k handles (each its own stream) render the strips of ranks 0..k-1 of the same
frames concurrently, so that one handle's launch sequence fills the gaps of another's -- the tails of a persistent
launch, and the fixed cost of every round of a deferred-walk sequence.  ms per frame for k = 1, 2, 3, 4 on the config 3
stand-in (default) or Cornell (SCENE=cornell); BATCH frames per launch (env, default 32)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
g = os.path.join(ROOT, "tests", "golden")
W, H = 1920, 1080
B = int(os.environ.get("BATCH", 32))
if os.environ.get("SCENE") == "cornell":
    arrays, spp = rt.SceneArrays.load(os.path.join(g, "cornell_scene.npz")), 8
else:
    arrays, spp = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                                   scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=3, device=0)), 16
for k in (1, 2, 3, 4):
    hs = [rt.RayTracer(0, W, H) for _ in range(k)]
    for h in hs:
        h.load_scene(arrays)
        h.set_option("batch_frames", B)
    best = 1e9
    for rep in range(3):
        p = rt.make_params(W, H, 4, spp, skybox=1, frames=rep * B)
        for h in hs:
            h.synchronize()
        t0 = time.perf_counter()
        for r, h in enumerate(hs):
            h.render_strips_frames(p, B, r, k)
        for h in hs:
            h.synchronize()
        if rep:
            best = min(best, (time.perf_counter() - t0) / B * 1e3)
    print(f"{k} handle(s) on one device: {best:.3f} ms per frame ({B} frames per launch)", flush=True)
    for h in hs:
        h.close()
