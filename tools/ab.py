#!/usr/bin/env python3
"""Interleaved A/B timing of kernel tuning options in ONE process (methodology:
several rounds, variants interleaved, median and min of the per-launch HIP-event
times).  Usage: python tools/ab.py "variant=1" "variant=0,waves=3072" ...
Each argument is one configuration: comma-separated option=value pairs
(kernel_variant|variant, persistent_waves|waves, plus any rt_set_option name).
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

ALIAS = {"variant": "kernel_variant", "blocks": "persistent_blocks", "lds": "lds_scene", "fb": "tile_feedback", "cull": "cull_roots", "pc": "pixel_cache", "ve": "vote_eighths", "vp": "vote_patience"}
UPLOAD_OPTS = {"tlas", "tlas_min", "forest"}
DEFAULTS = {"primary_table": 1, "tlas": 1, "tlas_min": 8, "forest": 1, "stack_wide": -1, "tile_feedback_period": 8, "kernel_variant": -1, "lds_scene": 1, "tile_feedback": 1, "cull_roots": -1, "pixel_cache": 1, "vote_eighths": -1, "vote_patience": -1}


def main():
    W, H, SPP, NB = 1920, 1080, 8, 4
    scene = os.environ.get("AB_SCENE", os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
    arrays = rt.SceneArrays.load(scene)
    tr = rt.RayTracer(0, W, H)
    tr.load_scene(arrays)
    configs = []
    for arg in sys.argv[1:] or ["variant=0"]:
        opts = {}
        for kv in arg.split(","):
            k, v = kv.split("=")
            opts[ALIAS.get(k, k)] = int(v)
        configs.append((arg, opts))
    rounds, per = int(os.environ.get("AB_ROUNDS", 5)), int(os.environ.get("AB_FRAMES", 4))
    times = {name: [] for name, _ in configs}
    last_upload_opts = set()
    for r in range(rounds + 1):
        for name, opts in configs:
            for k, v in DEFAULTS.items():
                try:
                    tr.set_option(k, v)
                except rt.RtError:
                    pass  # older experimental library without this option
            for k, v in opts.items():
                tr.set_option(k, v)
            if len(UPLOAD_OPTS & (set(opts) | last_upload_opts)) > 0:  # these take effect at upload time
                tr.update_buffers(arrays)
            last_upload_opts = UPLOAD_OPTS & set(opts)
            tr.reset_timing()
            for f in range(per):
                tr.render(rt.make_params(W, H, NB, SPP, skybox=1, frames=1 + f))
            st = tr.stats()
            if r > 0:  # round 0 is warm-up
                times[name].append(st.kernel_ms / st.launches)
    for name, _ in configs:
        t = times[name]
        print(f"{name:40s} median {statistics.median(t):8.3f} ms  min {min(t):8.3f} ms  ({len(t)} rounds x {per} frames)")


if __name__ == "__main__":
    main()
