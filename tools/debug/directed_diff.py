#!/usr/bin/env python3
"""Where do the kernels differ from the oracle on a directed family (tools/prune_directed.py)?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import prune_directed as pd  # noqa: E402
import ray_tracer_2_amd as rt  # noqa: E402
from oracle import oracle  # noqa: E402

want = sys.argv[1] if len(sys.argv) > 1 else "blades_rotated"
W, H = pd.W, pd.H
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
for name, arrays in pd.families():
    if name != want:
        continue
    p = rt.make_params(W, H, 1, 1, skybox=1, frames=0)
    ref, st = oracle.render(p, arrays)
    tr = rt.RayTracer(0, W, H)
    tr.load_scene(arrays)
    tr.set_camera(arrays.uniform.camera)
    for opts in ({"lds_scene": 1, "cross_prune": 0}, {"lds_scene": 0, "cross_prune": 0}, {"lds_scene": 0, "cross_prune": 0, "pixel_cache": 0},
                 {"lds_scene": 0, "cross_prune": 0, "primary_table": 0}, {"lds_scene": 0, "cross_prune": 0, "cull_roots": 0},
                 {"lds_scene": 0, "cross_prune": 0, "tlas": 0}, {"lds_scene": 0, "cross_prune": 1}, {"lds_scene": 1, "cross_prune": 1}):
        for k, v in opts.items():
            tr.set_option(k, v)
        if "tlas" in opts:
            tr.load_scene(arrays)
            tr.set_camera(arrays.uniform.camera)
        tr.render(p)
        got = tr.read_image(W, H)
        bad = np.argwhere(np.any(bits(got) != bits(ref), axis=-1))
        print(opts, "->", len(bad), "texels differ", tr.last_launch())
        for y, x in bad[:6]:
            print("   pixel", x, y, "gpu", got[y, x], "oracle", ref[y, x])
            rgba, rec = oracle.trace_pixel(p, arrays, int(x), int(y))
            print("      oracle transcript:", rec)
        for k in opts:
            tr.set_option(k, {"lds_scene": 1, "cross_prune": 0, "pixel_cache": 1, "primary_table": 1, "cull_roots": -1, "tlas": 1}[k])
        if "tlas" in opts:
            tr.load_scene(arrays)
            tr.set_camera(arrays.uniform.camera)
    # the counter kernels (never prune, re-intersect everything)
    tr.set_option("lds_scene", 0)
    tr.set_counters(True)
    tr.render(p)
    got = tr.read_image(W, H)
    bad = np.argwhere(np.any(bits(got) != bits(ref), axis=-1))
    print("counter kernels, global memory ->", len(bad), "texels differ")
    tr.set_counters(False)
    # debug views at the differing pixels
    for dbg in (1, 2, 5, 6):
        pdbg = rt.make_params(W, H, 1, 1, skybox=1, frames=0)
        pdbg.debug_flag = dbg
        pdbg.debug_scale = 100
        refd, _ = oracle.render(pdbg, arrays)
        tr.render(pdbg)
        gd = tr.read_image(W, H)
        print("debug view", dbg, "texels differing:", int(np.count_nonzero(np.any(bits(gd) != bits(refd), axis=-1))))
