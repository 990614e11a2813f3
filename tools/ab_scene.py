#!/usr/bin/env python3
"""Interleaved A/B timing of tuning options on one of the stand-in scenes (one process, several rounds,
median of HIP-event frame times):  python tools/ab_scene.py <n> <spp> "opt=val,opt=val" "..." ...
n = 0: sponza stand-in, n >= 1: dragon x n^2 in the Cornell box, n = -1: Cornell (config 2).
Env: AB_BATCH (frames per launch, default 4), AB_FRAMES (default 8), AB_ROUNDS (default 3), BS_W/BS_H/BS_BOUNCES."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

UPLOAD_OPTS = {"tlas", "tlas_min", "forest", "flat2"}
DEFAULTS = {"primary_table": 1, "tlas": 1, "tlas_min": 8, "forest": 1, "stack_wide": -1, "tile_feedback_period": 8, "kernel_variant": -1,
            "lds_scene": 1, "tile_feedback": 1, "cull_roots": -1, "pixel_cache": 1, "vote_eighths": -1, "vote_patience": -1, "lds_top": 0, "flat2": 1, "batch_tile_major": 1, "sort_rounds": -1,
            "memo_in_table": 1, "cross_prune": 0, "primary_hits": 1}


def main():
    n, spp = int(sys.argv[1]), int(sys.argv[2])
    g = os.path.join(ROOT, "tests", "golden")
    if n < 0:
        arrays = rt.SceneArrays.load(os.path.join(g, "cornell_scene.npz"))
    elif n == 0:
        arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(int(os.environ.get("BS_MESHES", 200)), detail=int(os.environ.get("BS_DETAIL", 1))))
    else:
        arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                                 scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n, device=0))
    W, H, NB = int(os.environ.get("BS_W", 1920)), int(os.environ.get("BS_H", 1080)), int(os.environ.get("BS_BOUNCES", 4))
    batch, frames, rounds = int(os.environ.get("AB_BATCH", 4)), int(os.environ.get("AB_FRAMES", 8)), int(os.environ.get("AB_ROUNDS", 3))
    tr = rt.RayTracer(0, W, H)
    tr.load_scene(arrays)
    configs = []
    for arg in sys.argv[3:] or [""]:
        configs.append((arg or "defaults", dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in arg.split(",") if kv)))
    times = {name: [] for name, _ in configs}
    last_upload = set()
    for r in range(rounds + 1):
        for name, opts in configs:
            for k, v in DEFAULTS.items():
                tr.set_option(k, v)
            for k, v in opts.items():
                tr.set_option(k, v)
            tr.set_option("batch_frames", max(1, batch))
            if UPLOAD_OPTS & (set(opts) | last_upload):
                tr.update_buffers(arrays)
            last_upload = UPLOAD_OPTS & set(opts)
            tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=0), max(2, batch))   # tile order for these options
            tr.reset_timing()
            tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=2), frames)
            st = tr.stats()
            if r > 0:
                times[name].append(st.kernel_ms / st.frames)
    for name, _ in configs:
        t = times[name]
        print(f"{name:50s} median {statistics.median(t):8.3f} ms/frame  min {min(t):8.3f}  ({len(t)} rounds x {frames} frames, {batch} per launch)", flush=True)


if __name__ == "__main__":
    main()
