#!/usr/bin/env python3
"""Estimate the fixed tail of the persistent kernel: same view at 1x, 2x, 4x the pixels."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, 3840, 2160)
tr.load_scene(arrays)
for (W, H) in ((960, 540), (1920, 1080), (2720, 1528), (3840, 2160)):
    for variant in (0, 1):
        tr.set_option("kernel_variant", variant)
        ts = []
        for r in range(4):
            tr.reset_timing()
            for f in range(3):
                tr.render(rt.make_params(W, H, 4, 8, frames=1 + f))
            st = tr.stats()
            if r:
                ts.append(st.kernel_ms / st.launches)
                rays = st.segments / st.launches
        t = statistics.median(ts)
        print(f"{W}x{H} variant {variant}: {t:.3f} ms, {rays / t / 1e3:.0f} Mrays/s, {t / (W * H) * 1e6:.3f} ns/pixel")
