#!/usr/bin/env python3
"""Cross-mesh pruning against the unpruned walk AT SCALE, on the GPU: every frame is rendered twice by the many-mesh product
kernels -- option cross_prune on (the default) and off (every mesh walked from an infinite distance, as the shader does) --
with a seed of its own (Params.frames = -f: a plain store, |frames| seeds, wgsl:475), and the two images are compared bit
for bit.  The hypothesis of DESIGN.md section 2.4 says they are equal; this counts the texels where they are not.

    python tools/prune_compare.py [frames=500] [scene=sponza340|sponza200|shear<seed>]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
which = sys.argv[2] if len(sys.argv) > 2 else "sponza340"
W, H = 1920, 1080
if which.startswith("sponza"):
    m = int(which[6:])
    arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(m, detail=8 if m >= 300 else 1))
else:   # a seeded random many-mesh scene in sheared local spaces (tools/fuzz_parity.py)
    os.environ["FUZZ_MANY"] = "1"
    from test_gpu_scenes import _random_scene
    seed = int(which[5:])
    arrays = _random_scene(rt, seed, many=True)
    import importlib.util
    spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    src = open(os.path.join(ROOT, "tools", "fuzz_parity.py")).read()
    ns = {"np": np, "rt": rt}
    exec(src[src.index("def sheared"):src.index("first = int(")], ns)
    arrays = ns["sheared"](arrays, seed)
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
assert tr.last_launch is not None
rays = bad_texels = bad_frames = 0
t0 = time.time()
for f in range(n):
    p = rt.make_params(W, H, 4, 8, skybox=1, frames=-f)
    tr.reset_timing()
    tr.set_option("cross_prune", 1)
    tr.render(p)
    a = tr.read_image(W, H).view(np.uint32)
    rays += tr.stats().segments
    tr.set_option("cross_prune", 0)
    tr.render(p)
    b = tr.read_image(W, H).view(np.uint32)
    d = int((a != b).any(-1).sum())
    if d:
        bad_frames += 1
        bad_texels += d
        print(f"frame {f}: {d} texels differ", flush=True)
    if f % 100 == 99:
        print(f"... {f + 1} frames, {rays / 1e9:.1f} G rays, {bad_texels} differing texels, {time.time() - t0:.0f} s", flush=True)
assert tr.last_launch()["many_mesh"]
print(f"{which}: {n} frames of {W}x{H}, 8 spp, 4 bounces = {rays / 1e9:.2f} G rays rendered with and without cross-mesh pruning: "
      f"{bad_texels} differing texels in {bad_frames} frames")
