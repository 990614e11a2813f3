#!/usr/bin/env python3
"""Independent single frames (a moving camera: every frame starts a new accumulation) pipelined across TWO handles on one
device: each handle has its own stream and image, so frame k+1's launch fills the CUs that frame k's draining waves free.
Reports ms per frame for one handle and for two handles used alternately (config 2 frame, one launch per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
W, H, N = 1920, 1080, 200
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
hs = [rt.RayTracer(0, W, H) for _ in range(3)]
for h in hs:
    h.load_scene(arrays)
    h.render(rt.make_params(W, H, 4, 8, skybox=1, frames=0))
    h.synchronize()
for k in (1, 2, 3):
    best = 1e9
    for rep in range(3):
        for h in hs[:k]:
            h.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            hs[i % k].render(rt.make_params(W, H, 4, 8, skybox=1, frames=0))
        for h in hs[:k]:
            h.synchronize()
        best = min(best, (time.perf_counter() - t0) / N * 1e3)
    print(f"{k} handle(s): {best:.3f} ms per frame", flush=True)
