#!/usr/bin/env python3
"""Two batches of BATCH (env, default 8) frames of a dragon stand-in (env SUBDIV 3, W 1920, H 1080, SPP 16, BOUNCES 4) with
sort_rounds = ROUNDS (env, default 10), for rocprofv3 --kernel-trace: the per-launch durations show where a deferred-walk
sequence spends its time (tools/rounds_timeline.py prints them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
g = os.path.join(ROOT, "tests", "golden")
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                         scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=int(os.environ.get("SUBDIV", 3)), device=0 if int(os.environ.get("SUBDIV", 3)) > 3 else None))
W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
B = int(os.environ.get("BATCH", 8))
tr.set_option("batch_frames", B)
tr.set_option("sort_rounds", int(os.environ.get("ROUNDS", 10)))
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        tr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p = rt.make_params(W, H, int(os.environ.get("BOUNCES", 4)), int(os.environ.get("SPP", 16)), skybox=1, frames=0)
tr.render_frames(p, B)
tr.synchronize()
p.frames = B
tr.render_frames(p, B)
tr.synchronize()
