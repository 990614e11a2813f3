#!/usr/bin/env python3
"""Two batches of BATCH (env, default 8) frames of the config 3 stand-in with sort_rounds = ROUNDS (env, default 10), for
rocprofv3 --kernel-trace: the per-launch durations show where a sequence of class launches spends its time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
g = os.path.join(ROOT, "tests", "golden")
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                         scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=3))
W, H = 1920, 1080
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
B = int(os.environ.get("BATCH", 8))
tr.set_option("batch_frames", B)
tr.set_option("sort_rounds", int(os.environ.get("ROUNDS", 10)))
p = rt.make_params(W, H, 4, 16, skybox=1, frames=0)
tr.render_frames(p, B)
tr.synchronize()
p.frames = B
tr.render_frames(p, B)
tr.synchronize()
