#!/usr/bin/env python3
"""Soak: tens of thousands of one-frame calls through every host-side path added in round 4 (pipelined frames, frames
rendered ahead, slot tables of a moving camera, snapshots, strip shares, size changes), device memory watched, the final
accumulation compared with a plain one-launch-per-frame run of the same frames."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: its HIP runtime serves both)
import ray_tracer_2_amd as rt  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W, H = 320, 180
if len(sys.argv) > 2 and sys.argv[2] == "many":   # a many-mesh textured scene, read from global memory
    from ray_tracer_2_amd import scenes
    arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(200))
else:
    arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
cam_t = type(arrays.uniform.camera)


def free_mb():
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def script(tr, n, fancy):
    rng = np.random.RandomState(7)
    f = 0
    tr.write_image(np.zeros((H, W, 4), np.float32))
    low = None
    for k in range(n):
        r = rng.rand()
        if r < 0.01:                                  # the camera moves for a few frames, then comes back
            for m in range(1 + int(rng.randint(4))):
                c = cam_t.from_buffer_copy(bytes(arrays.uniform.camera))
                c.cam_to_world[3][0] += 0.01 * (m + 1)
                tr.set_camera(c)
                tr.render(rt.make_params(W, H, 3, 2, skybox=1, frames=0))
            tr.set_camera(arrays.uniform.camera)
            tr.write_image(np.zeros((H, W, 4), np.float32))
            f = 0
        elif r < 0.015:                               # another size for a while
            for m in range(3):
                tr.render(rt.make_params(96, 54, 2, 1, skybox=1, frames=m))
            tr.write_image(np.zeros((H, W, 4), np.float32))
            f = 0
        elif r < 0.02:                                # a strip share in between
            tr.render_strips(rt.make_params(W, H, 3, 2, skybox=1, frames=0), 1, 4)
            tr.write_image(np.zeros((H, W, 4), np.float32))
            f = 0
        tr.render(rt.make_params(W, H, 3, 2, skybox=1, frames=f))
        f += 1
        if fancy and r > 0.97:
            tr.snapshot_image(W, H)
            tr.read_snapshot(W, H)
        if fancy and r > 0.995:
            tr.synchronize()                          # (a host that waits now and then)
        if k % 2000 == 1999:
            tr.synchronize()
            m = free_mb()
            low = m if low is None else min(low, m)
            print(f"  {k + 1} calls, free device memory {m:.0f} MiB", flush=True)
    return tr.read_image(W, H).copy(), low


t0 = time.time()
a = rt.RayTracer(0, W, H)
a.load_scene(arrays)
a.set_option("frame_ahead", 0)
a.set_option("pipeline", 0)
a.set_option("primary_per_slot", 0)
want, _ = script(a, N, False)
a.close()
b = rt.RayTracer(0, W, H)
b.load_scene(arrays)
start = free_mb()
got, low = script(b, N, True)
print(f"{N} calls per run, {time.time() - t0:.0f} s; free memory at the start {start:.0f} MiB, lowest seen {low:.0f} MiB")
same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
print("final accumulation bit-identical to the plain run:", same)
sys.exit(0 if same and start - low < 600 else 1)
