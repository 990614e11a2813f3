#!/usr/bin/env python3
"""When do the persistent waves finish?  (diagnostic build; s_memrealtime = 100 MHz)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subprocess  # noqa: E402

SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x_wt.so")
if "--build-only" in sys.argv or not os.path.exists(SO):
    subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "wt", "-DRT_WAVE_TIMES=1"], check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
import ray_tracer_2_amd.lib as lib  # noqa: E402
lib.LIB_PATH = SO
import ray_tracer_2_amd as rt  # noqa: E402

W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
L = rt.load()
buf = (C.c_uint64 * (2 * 8192))()
for fb in (0, 1):
    tr.set_option("tile_feedback", fb)
    for f in range(3):
        tr.render(rt.make_params(W, H, 4, 8, frames=f))
    L.rt_diag_wave_times(tr._h, buf)
    t = np.array(buf, dtype=np.uint64).reshape(-1, 2)[:4096].astype(np.float64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # microseconds
    print(f"tile_feedback={fb}: waves start within {start.max():.0f} us; ends: p10 {np.percentile(end, 10):.0f}  p50 {np.percentile(end, 50):.0f} "
          f"p90 {np.percentile(end, 90):.0f}  p99 {np.percentile(end, 99):.0f}  max {end.max():.0f} us")
    busy = (end - start).sum() / (end.max() * len(end))
    print(f"   wave residency = {busy:.1%} of (waves x kernel time)")
