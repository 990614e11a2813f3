#!/usr/bin/env python3
"""The hypothesis behind cross-mesh pruning, measured on the CPU with the oracle: over every
triangle test that reports a hit, how far the slab test's entry distance of the triangle's LEAF box (the boxes above
it are entered no later) can lie BEYOND the hit's own parameter t.  Exact arithmetic: never.  The pruning is exact as
long as entry <= 1.125 t for the hits that matter.

    python tools/prune_census.py [rows=270]      (stand-ins of configs 4 and 3, evenly spaced rows of the 1920x1080 frame)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from oracle import oracle  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 270
W, H = 1920, 1080
g = os.path.join(ROOT, "tests", "golden")
cases = [
    ("config 4 stand-in at sponza.obj's size (340 meshes x 768 triangles)", lambda: rt.SceneArrays.from_scene(scenes.sponza_standin(340, detail=8)), 8, 4),
    ("config 4 stand-in (200 meshes x 12 triangles)", lambda: rt.SceneArrays.from_scene(scenes.sponza_standin(200)), 8, 4),
    ("config 3 stand-in (dragon.obj x 9 in the Cornell box)",
     lambda: rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                             scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=3)), 16, 4),
    ("config 2 (Cornell box)", lambda: rt.SceneArrays.load(os.path.join(g, "cornell_scene.npz")), 8, 4),
]
rows = np.unique((np.arange(rows_n) * H // rows_n)).astype(np.uint32)
for name, make, spp, nb in cases:
    a = make()
    oracle.census(True)
    t0 = time.time()
    segs = 0
    for f in range(2):
        _, st = oracle.render(rt.make_params(W, H, nb, spp, skybox=1, frames=f), a, image=np.zeros((H, W, 4), np.float32), rows=rows)
        segs += st.segments
    c = oracle.census(False)
    print(f"{name}: {segs} rays, {int(c['hits'])} triangle hits in {time.time() - t0:.0f} s")
    print(f"   leaf-box entry beyond the hit's t: {int(c['entry_gt_t'])} ({c['entry_gt_t'] / max(c['hits'], 1):.2e} of the hits); "
          f"by > 1e-6: {int(c['gt_1e-6'])}, > 1e-4: {int(c['gt_1e-4'])}, > 1 %: {int(c['gt_1pct'])}, > 12.5 %: {int(c['gt_12.5pct'])}; "
          f"largest entry / t = {c['max_ratio']:.9f}", flush=True)
