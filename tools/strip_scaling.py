#!/usr/bin/env python3
"""Compute part of the multi-GPU strip split, measured on ONE GPU: rank 0's share of the config-2
frame (8-row strips dealt round-robin) for world = 1, 2, 4, 8: one launch per frame with and without the
pipeline, with frames overlapped (rt_render_strips_frames, 32 frames per launch), and one rt_render_strips CALL per frame
under the default options (option frame_ahead: a call that continues an accumulation renders the next frames with its
own; every call still leaves its frame in the image); wall time per frame.  Predicts the compute-only scaling; the
gather (<= 4.2 MB per rank per batch over xGMI) is not included."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

import time  # noqa: E402

W, H, N = 1920, 1080, 64
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
base = {}
# (mode, frames per launch, option pipeline): wall time per frame -- pipelined launches overlap, their event times do not add up
for mode, batch, pipe in (("one launch per frame, no pipeline", 1, 0), ("one launch per frame, four frames in flight", 1, 4),
                          ("one launch per frame, 4 frames (7 from world 4)", 1, -2), ("32 frames per launch", 32, -1)):
    tr.set_option("batch_frames", batch)
    for world in (1, 2, 4, 8):
        tr.set_option("pipeline", (7 if world >= 4 else 4) if pipe == -2 else pipe)
        ts = []
        for r in range(4):
            tr.synchronize()
            t0 = time.perf_counter()
            tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
            tr.synchronize()
            if r:
                ts.append((time.perf_counter() - t0) / N * 1e3)
        t = statistics.median(ts)
        if world == 1:
            base[mode] = t
        print(f"{mode:48s} world {world}: {t:.3f} ms per frame for rank 0's share -> compute-only speed-up {base[mode] / t:.2f}x", flush=True)
# one CALL per frame (the reference's mode: app.rs:44-53 advances Params.frames, one render per redraw)
tr.set_option("batch_frames", 32)
tr.set_option("pipeline", -1)
N = 256
for mode, ahead in (("one call per frame, frame_ahead off", 0), ("one call per frame, default options", -1)):
    tr.set_option("frame_ahead", ahead)
    for world in (1, 2, 4, 8):
        ts = []
        f = 0   # (one accumulation that goes on through the repetitions: the steady state of a standing camera)
        for r in range(5):
            tr.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                tr.render_strips(rt.make_params(W, H, 4, 8, skybox=1, frames=f), 0, world)
                f += 1
            tr.synchronize()
            if r:
                ts.append((time.perf_counter() - t0) / N * 1e3)
        t = statistics.median(ts)
        if world == 1:
            base[mode] = t
        print(f"{mode:48s} world {world}: {t:.3f} ms per frame for rank 0's share -> compute-only speed-up {base[mode] / t:.2f}x", flush=True)
