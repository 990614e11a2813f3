#!/usr/bin/env python3
"""Compute part of the multi-GPU strip split, measured on ONE GPU: rank 0's share of the config-2
frame (8-row strips dealt round-robin) for world = 1, 2, 4, 8, one launch per frame and with frames
overlapped (rt_render_strips_frames, 16 frames per launch).  Predicts the compute-only scaling; the
gather (<= 4.2 MB per rank per batch over xGMI) is not included."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H, N = 1920, 1080, 64
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
base = {}
for batch in (1, 16):
    tr.set_option("batch_frames", batch)
    for world in (1, 2, 4, 8):
        ts = []
        for r in range(4):
            tr.reset_timing()
            tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), N, 0, world)
            st = tr.stats()
            if r:
                ts.append(st.kernel_ms / st.frames)
        t = statistics.median(ts)
        if world == 1:
            base[batch] = t
        mode = "one launch per frame" if batch == 1 else f"{batch} frames per launch"
        print(f"{mode:22s} world {world}: {t:.3f} ms per frame for rank 0's share -> compute-only speed-up {base[batch] / t:.2f}x "
              f"(vs the un-overlapped 1-GPU frame: {base[1] / t:.2f}x)", flush=True)
