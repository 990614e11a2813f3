#!/usr/bin/env python3
"""Deferred walks (option sort_rounds) against the plain kernels on the config 3 stand-in, bit for bit: batches and single
frames, several round counts, the stats counters, strips."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
g = os.path.join(ROOT, "tests", "golden")
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                         scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=3, device=0))
W, H = 480, 270
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
p = rt.make_params(W, H, 4, 4, skybox=1, frames=0)
bad = 0
def run(R, counters, n):
    tr.set_option("sort_rounds", R)
    tr.set_counters(counters)
    tr.write_image(np.zeros((H, W, 4), np.float32))
    tr.reset_timing()
    if n == 1:
        tr.render(p)
    else:
        tr.render_frames(p, n)
    st = tr.stats()
    return tr.read_image(W, H).copy(), (st.segments, st.node_tests, st.triangle_tests)
for counters in (False, True):
    for n in (1, 3):
        ref, cref = run(0, counters, n)
        for R in (1, 2, 5, 20):
            img, c = run(R, counters, n)
            ok = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) and c == cref
            bad += not ok
            print(f"counters {counters} frames {n} rounds {R}: {'ok' if ok else 'MISMATCH'} {c} {cref if not ok else ''}", flush=True)
tr.set_counters(False)
# strips: rank r of 3 with rounds == without
for rank in range(3):
    outs = []
    for R in (0, 4):
        tr.set_option("sort_rounds", R)
        tr.write_image(np.zeros((H, W, 4), np.float32))
        tr.render_strips_frames(p, 3, rank, 3)
        outs.append(tr.read_image(W, H).copy())
    ok = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    bad += not ok
    print(f"strips rank {rank}/3: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
