#!/usr/bin/env python3
"""Per-GPU kernel time of the strip split, measured on ONE GPU: rank 0's share
of the config-2 frame for world = 1, 2, 4, 8 (predicts the compute part of the
multi-GPU scaling; the gather is not included)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
base = None
for variant in (0, 1):
    tr.set_option("kernel_variant", variant)
    for world in (1, 2, 4, 8):
        ts = []
        for r in range(5):
            tr.reset_timing()
            for f in range(4):
                tr.render_strips(rt.make_params(W, H, 4, 8, frames=1 + f), 0, world)
            st = tr.stats()
            if r:
                ts.append(st.kernel_ms / st.launches)
        t = statistics.median(ts)
        if world == 1:
            base = t
        print(f"variant {variant} world {world}: {t:.3f} ms per rank-0 share -> compute-only speed-up {base / t:.2f}x")
