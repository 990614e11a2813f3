#!/usr/bin/env python3
"""A camera that moves every frame (frames = 0 each call: the accumulation restarts, the primary table follows the
camera), config 2's scene and size, host running ahead: ms per frame with the shared primary table rewritten behind a
barrier (primary_per_slot = 0) and with a table per pipeline slot (1, the default)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
cam_t = type(arrays.uniform.camera)
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
for per_slot in (0, 1, 2):
    tr.set_option("primary_per_slot", per_slot & 1)
    tr.set_option("primary_table", 0 if per_slot == 2 else 1)   # (2: no table at all -- every pixel computes its own memo)
    res = []
    for rep in range(4):
        tr.synchronize()
        t0 = time.perf_counter()
        n = 96
        for f in range(n):
            cam = cam_t.from_buffer_copy(bytes(arrays.uniform.camera))
            cam.cam_to_world[3][0] += 0.002 * (f + 1 + rep * n)
            tr.set_camera(cam)
            tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=0))
        tr.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
    print(f"{'no primary table' if per_slot == 2 else f'primary_per_slot {per_slot}'}: {statistics.median(res[1:]):.3f} ms per frame of a moving camera", flush=True)
tr.set_option("primary_table", 1)
tr.set_camera(arrays.uniform.camera)
