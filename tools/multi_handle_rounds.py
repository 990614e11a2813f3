#!/usr/bin/env python3
"""Would deferred walks pay for single frames if several frames were in flight?  N handles on one device render independent
frames of a dragon stand-in concurrently, with and without deferred walks (sort_rounds forced); aggregate wall time per frame.
    python tools/multi_handle_rounds.py [subdivide 3] [spp 16]        (answer, round 3: no -- tools/experiments/README.md)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
g = os.path.join(ROOT, "tests", "golden")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")), scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n, device=0 if n > 3 else None))
W, H, SPP = 1920, 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 16
for nh, rounds, pipe in ((1, 0, 4), (4, 0, 0), (4, 1, 0), (4, 2, 0), (4, 3, 0), (3, 2, 0), (6, 2, 0)):
    hs = []
    for k in range(nh):
        t = rt.RayTracer(0, W, H)
        t.set_option("sort_rounds", rounds)
        t.set_option("pipeline", pipe)
        t.load_scene(arrays)
        hs.append(t)
    def run(f0, nf):
        for f in range(nf):
            for t in hs:
                t.render(rt.make_params(W, H, 4, SPP, skybox=1, frames=0 if nh > 1 else f0 + f))
    run(0, 3)
    for t in hs: t.synchronize()
    t0 = time.perf_counter()
    NF = 6
    run(3, NF)
    for t in hs: t.synchronize()
    dt = (time.perf_counter() - t0) * 1e3 / (NF * nh)
    print(f"handles {nh} sort_rounds {rounds} pipeline {pipe}: {dt:.3f} ms per frame", hs[0].last_launch()["deferred_walks"], flush=True)
    for t in hs: t.close()
