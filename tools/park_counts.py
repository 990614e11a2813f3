#!/usr/bin/env python3
"""Pixels parked in front of each round of a deferred-walk sequence (config 3 stand-in by default): how fast the parked
population decays -- the figure that decides how many rounds pay (rt_api.hip: n_rounds).
    python tools/park_counts.py [subdivide=3] [spp=16] [batch=32] [rounds=-1]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else -1
g = os.path.join(ROOT, "tests", "golden")
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                         scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")),
                                                         subdivide=n, device=0 if n > 3 else None))
W, H = 1920, 1080
tr = rt.RayTracer(0, W, H, lib=rt.load_test())   # (rt_test_read_wavefront: the test library)
tr.set_option("batch_frames", batch)
tr.set_option("sort_rounds", rounds)
tr.load_scene(arrays)
for rep in range(2):
    tr.render_frames(rt.make_params(W, H, 4, spp, skybox=1, frames=rep * batch), batch)
tr.synchronize()
out = np.zeros(72, np.uint32)
rc = tr._L.rt_test_read_wavefront(tr._h, 4, out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
assert rc == 0, rc
counts = [int(c) for c in out if c]
print(f"dragon x{n * n}: {W}x{H}, {spp} spp, {batch} frames per launch: {W * H * batch} pixels; parked in front of round k:")
for k, c in enumerate(counts):
    print(f"  round {k + 1}: {c:9d}  ({c / (W * H * batch):.3f} of the pixels{'' if k == 0 else f', {c / counts[k - 1]:.3f} of the round before'})")

# how many of the last round's walks hit the mesh at all (plane 13: tri = ~0 means no hit)
last = len(counts) - 1
n = counts[last]
blocks = (n + 63) // 64
raw = np.zeros(blocks * 14 * 64 * 4, np.float32)
rc = tr._L.rt_test_read_wavefront(tr._h, 5 + (last & 1), raw.ctypes.data_as(ctypes.c_void_p), raw.nbytes)
assert rc == 0, rc
rec = raw.view(np.uint32).reshape(blocks, 14, 64, 4)
tri = rec[:, 13, :, 3].reshape(-1)[:n]
closest = rec[:, 10, :, 0].reshape(-1)[:n].view(np.float32)
t = rec[:, 13, :, 0].reshape(-1)[:n].view(np.float32)
hit = tri != 0xffffffff
print(f"last round: {n} walks, {hit.mean():.3f} of them hit the deferred mesh")

if os.environ.get("PARK_SAMPLE"):   # a sample of the last round's rays for offline analysis (tools/park_probe_model.py)
    k = min(n, 40000)
    idx = np.random.RandomState(1).choice(n, k, replace=False)
    p2 = rec[:, 2, :, :].reshape(-1, 4)[:n][idx].view(np.float32)
    p3 = rec[:, 3, :, :].reshape(-1, 4)[:n][idx].view(np.float32)
    np.savez_compressed(os.environ["PARK_SAMPLE"], ro=p2[:, :3], rd=np.concatenate([p2[:, 3:4], p3[:, :2]], axis=1),
                        hit=hit[idx], t=t[idx], closest=closest[idx])
    print("sample written:", os.environ["PARK_SAMPLE"])
