#!/usr/bin/env python3
"""Per-rank kernel time at world = 8 for different persistent grid sizes."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
tr.set_option("kernel_variant", 0)
for world in (8, 4):
    for blocks in (256, 384, 512, 640, 768, 896, 1024):
        tr.set_option("persistent_blocks", blocks)
        ts = []
        for r in range(4):
            tr.reset_timing()
            for f in range(4):
                tr.render_strips(rt.make_params(W, H, 4, 8, frames=1 + f), 0, world)
            st = tr.stats()
            if r:
                ts.append(st.kernel_ms / st.launches)
        print(f"world {world} blocks {blocks}: {statistics.median(ts):.3f} ms")
