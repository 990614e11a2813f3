#!/usr/bin/env python3
"""Rank 0's share of the config-2 frame at world = 1, 2, 4, 8 in back-to-back batched launches of 2 .. 32 frames (no
pipeline): how many frames a launch needs before the persistent kernel's lane refill pays -- the figures behind the
automatic depth of option frame_ahead (rt_api.hip: ahead_depth).  profiles/r04_strip_batch_sweep.txt."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
W, H, N = 1920, 1080, 64
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
for batch in (2, 4, 8, 16, 32):
    tr.set_option("batch_frames", batch)
    tr.set_option("pipeline", 0)
    for world in (1, 2, 4, 8):
        ts = []
        for r in range(4):
            tr.synchronize()
            t0 = time.perf_counter()
            tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
            tr.synchronize()
            if r:
                ts.append((time.perf_counter() - t0) / N * 1e3)
        print(f"batch {batch:2d} world {world}: {statistics.median(ts):.4f} ms/frame", flush=True)
