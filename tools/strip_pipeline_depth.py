#!/usr/bin/env python3
"""One launch per frame on rank 0's share of the config-2 frame for world = 1, 2, 4, 8 (measured on one GPU): wall time per
frame against the frames in flight (option pipeline).   GPU_MAX_HW_QUEUES=16 python tools/strip_pipeline_depth.py"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H, N = 1920, 1080, 96
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)


def run(world):
    ts = []
    for r in range(4):
        tr.synchronize()
        t0 = time.perf_counter()
        tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
        tr.synchronize()
        if r:
            ts.append((time.perf_counter() - t0) / N * 1e3)
    return statistics.median(ts)


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for world in (1, 2, 4, 8):
    tr.set_option("batch_frames", 32)
    line = [f"world {world}: 32 frames per launch {run(world):.4f} |"]
    tr.set_option("batch_frames", 1)
    for depth in (0, 2, 3, 4, 5, 6, 7, 8):
        tr.set_option("pipeline", depth)
        line.append(f"{depth}: {run(world):.4f}")
    print(" ".join(line), flush=True)
