#!/usr/bin/env python3
"""What it costs to get a frame to the host (rt_read_image, 1920x1080 RGBA32F = 33 MB): into pageable memory (numpy),
and with the read of frame k under the rendering of frame k + 1 (rt_snapshot_image / rt_read_snapshot)."""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=0))
tr.synchronize()
L = tr._L
nbytes = W * H * 16


def timed(fn, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


out = np.empty((H, W, 4), np.float32)
out[:] = 0   # (touch the pages)
t = timed(lambda: L.rt_read_image(tr._h, out.ctypes.data, nbytes))
print(f"rt_read_image into pageable memory: {t:.3f} ms ({nbytes / t / 1e6:.1f} GB/s)")


def loop(overlap, n=64):
    tr.synchronize()
    t0 = time.perf_counter()
    tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=1))
    if overlap:
        tr.snapshot_image(W, H)
    for f in range(1, n):
        if overlap:
            tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=1 + f))
            L.rt_read_snapshot(tr._h, out.ctypes.data, nbytes)       # frame f - 1, while frame f renders
            tr.snapshot_image(W, H)
        else:
            L.rt_read_image(tr._h, out.ctypes.data, nbytes)
            tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=1 + f))
    if overlap:
        L.rt_read_snapshot(tr._h, out.ctypes.data, nbytes)
    else:
        L.rt_read_image(tr._h, out.ctypes.data, nbytes)
    return (time.perf_counter() - t0) / n * 1e3


for ahead in (-1, 0):
    tr.set_option("frame_ahead", ahead)
    for name, ov in (("render, rt_read_image, render, ...", False),
                     ("render k + 1, rt_read_snapshot (frame k), rt_snapshot_image, ...", True)):
        loop(ov, 8)
        print(f"frame_ahead {ahead:2d}: {name}: {statistics.median([loop(ov) for _ in range(3)]):.3f} ms per shown frame")
