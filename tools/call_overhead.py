#!/usr/bin/env python3
"""Host time of one rt_render call (the API calls behind it: stream waits, memsets, launches, event records), measured on
frames so small that the GPU is never the limit (16 x 16, 1 spp): the plain launch, the pipelined launch, and a call whose
frame was rendered ahead (one blend launch).  The answer to "would a hipGraph pay?": a frame of config 2 lasts 1.13 ms, a
strip share of eight ranks 0.16 ms."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

W = H = 16
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
N = 4000
for name, opts in (("plain launch (pipeline 0, frame_ahead 0)", {"pipeline": 0, "frame_ahead": 0}),
                   ("pipelined launch (pipeline 4, every frame through it, frame_ahead 0)", {"pipeline": 4, "pipeline_when_idle": 1, "frame_ahead": 0}),
                   ("frames rendered ahead (frame_ahead 32: 31 of 32 calls are one blend launch)", {"pipeline": 1, "frame_ahead": 32})):
    for k, v in opts.items():
        tr.set_option(k, v)
    res = []
    for rep in range(4):
        tr.synchronize()
        t0 = time.perf_counter()
        for f in range(N):
            tr.render(rt.make_params(W, H, 1, 1, skybox=1, frames=f))
        t1 = time.perf_counter()
        tr.synchronize()
        res.append((t1 - t0) / N * 1e6)
    print(f"{name}: {statistics.median(res[1:]):.1f} us of host time per call (Python + ctypes included)", flush=True)
    tr.set_option("pipeline", 1)
    tr.set_option("pipeline_when_idle", 0)
    tr.set_option("frame_ahead", -1)
