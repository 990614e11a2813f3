#!/usr/bin/env python3
"""A host that renders, WAITS for the frame, renders the next (the reference's present loop without run-ahead), config 2:
ms per frame over one long accumulation, with and without the frames rendered ahead (option frame_ahead)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
for ahead in (0, -1):
    tr.set_option("frame_ahead", ahead)
    f = 0
    res = []
    for rep in range(4):
        t0 = time.perf_counter()
        worst = 0.0
        for _ in range(256):
            t1 = time.perf_counter()
            tr.render(rt.make_params(W, H, 4, 8, skybox=1, frames=f))
            tr.synchronize()
            worst = max(worst, time.perf_counter() - t1)
            f += 1
        res.append(((time.perf_counter() - t0) / 256 * 1e3, worst * 1e3))
    print(f"frame_ahead {ahead:2d}: {statistics.median(r[0] for r in res[1:]):.3f} ms per frame in the steady state "
          f"(first 256 frames, from cold: {res[0][0]:.3f}); longest wait for one frame {max(r[1] for r in res[1:]):.2f} ms")
