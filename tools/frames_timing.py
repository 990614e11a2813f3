#!/usr/bin/env python3
"""Config 2 (Cornell 1920x1080, 8 spp, 4 bounces): ms per frame of sequential rt_render calls against
rt_render_frames at several batch sizes (wall clock around a synchronised region, and HIP-event time).
    python tools/frames_timing.py [world]      # world > 1: rank 0's strips only (compute part of the split)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
N = 64


def run(batch):
    best = None
    for rep in range(4):
        tr.synchronize()
        tr.reset_timing()
        t0 = time.perf_counter()
        if batch == 0:
            for f in range(N):
                tr.render_strips(rt.make_params(W, H, 4, 8, skybox=1, frames=1 + f), 0, world)
        else:
            tr.set_option("batch_frames", batch)
            tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
        tr.synchronize()
        wall = (time.perf_counter() - t0) / N * 1e3
        st = tr.stats()
        ev = st.kernel_ms / st.frames
        if rep and (best is None or wall < best[0]):
            best = (wall, ev, st.segments / st.frames)
    return best


base = None
for batch in (0, 2, 4, 8, 16, 32):
    wall, ev, rays = run(batch)
    if base is None:
        base = wall
    print(f"world {world} batch {batch:2d}: {wall:.3f} ms/frame wall, {ev:.3f} ms/frame in kernels, "
          f"{rays / wall / 1e3:.0f} Mrays/s, x{base / wall:.2f}", flush=True)
