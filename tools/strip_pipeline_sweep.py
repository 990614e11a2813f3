#!/usr/bin/env python3
"""One launch per frame on a rank's share of the config-2 frame (8-way strip split, measured on one GPU): how the grid of
the persistent kernel and the pipeline depth should be sized when a frame no longer fills the machine (259 K pixels of
rank 0 for 328 K resident lanes).   python tools/strip_pipeline_sweep.py [world=8]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H, N = 1920, 1080, 96
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
tr.set_option("batch_frames", 1)


def run():
    ts = []
    for r in range(4):
        tr.synchronize()
        t0 = time.perf_counter()
        tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
        tr.synchronize()
        if r:
            ts.append((time.perf_counter() - t0) / N * 1e3)
    return statistics.median(ts)


tr.set_option("batch_frames", 32)
tr.set_option("pipeline", 4)
print(f"world {world}: 32 frames per launch {run():.4f} ms per frame", flush=True)
tr.set_option("batch_frames", 1)
for variant in (-1, 0, 1):
    for blocks in (1280, 960, 640, 512, 427, 320, 256, 160):
        if variant == 1 and blocks != 1280:
            continue
        for depth in (2, 3, 4):
            tr.set_option("kernel_variant", variant)
            tr.set_option("persistent_blocks", blocks)
            tr.set_option("pipeline", depth)
            print(f"variant {variant:2d} persistent_blocks {blocks:5d} pipeline {depth}: {run():.4f} ms per frame  {tr.last_launch()}", flush=True)
