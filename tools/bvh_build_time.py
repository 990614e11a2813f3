#!/usr/bin/env python3
"""Host vs GPU-assisted BVH build time for dragon.obj split n x n."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
sc = rt.Scene()
for _label, v, idx, _t, _m in scenes.load_raw_meshes(os.path.join(ROOT, "tests", "golden", "dragon_raw.npz")):
    sc.add_mesh_from_data(v, idx)
sc.subdivide_meshes(n)
out = {}
for name, kw in (("gpu (first call)", dict(device=0)), ("gpu", dict(device=0)), ("host", {})):
    t0 = time.perf_counter()
    sc.build(**kw)
    dt = time.perf_counter() - t0
    a = rt.SceneArrays.from_scene(sc)
    out[name] = (a.nodes.tobytes(), a.triangles.tobytes())
    print(f"{name:18s} {dt:8.2f} s   {a.triangles.shape[0]} triangles, {a.nodes.shape[0]} nodes")
print("identical:", out["gpu"] == out["host"])
