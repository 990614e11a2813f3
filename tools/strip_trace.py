#!/usr/bin/env python3
"""Workload for a kernel trace of the per-frame pipeline on a rank's strip share:
    rocprofv3 --kernel-trace -d gpurun_out/trace -- python3 tools/strip_trace.py [world=8] [depth=4] [frames=48]
and, with --analyse <csv>, the steady-state summary of such a trace (render / blend durations, launches in flight)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def analyse(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("kernel_name")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    render = [(s, e) for s, e, n in rows if "rt_render" in n]
    blend = [(s, e) for s, e, n in rows if "rt_blend" in n]
    render, blend = render[len(render) // 4:], blend[len(blend) // 4:]   # steady state
    span = (max(e for _, e in render) - render[0][0]) / 1e3
    print(f"{len(render)} render launches over {span:.0f} us: {span / len(render):.1f} us per frame")
    print(f"render kernel duration: mean {sum(e - s for s, e in render) / len(render) / 1e3:.1f} us, "
          f"min {min(e - s for s, e in render) / 1e3:.1f}, max {max(e - s for s, e in render) / 1e3:.1f}")
    print(f"blend kernel duration: mean {sum(e - s for s, e in blend) / max(len(blend), 1) / 1e3:.1f} us")
    busy = sum(e - s for s, e in render) / 1e3
    print(f"render launches in flight on average: {busy / span:.2f}")
    gaps = [(blend[i + 1][0] - blend[i][1]) / 1e3 for i in range(len(blend) - 1)]
    if gaps:
        print(f"gap between consecutive blends: mean {sum(gaps) / len(gaps):.1f} us")
    lag = []
    bi = 0
    for s, e in render:   # render end -> its blend's start (the next blend that starts after the render ended)
        while bi < len(blend) and blend[bi][0] < e:
            bi += 1
        if bi < len(blend):
            lag.append((blend[bi][0] - e) / 1e3)
    if lag:
        print(f"render end -> next blend start: mean {sum(lag) / len(lag):.1f} us")


if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    analyse(sys.argv[2])
    sys.exit(0)

import ray_tracer_2_amd as rt  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 48
W, H = 1920, 1080
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
tr.set_option("batch_frames", 1)
tr.set_option("pipeline", depth)
for rep in range(2):
    tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), frames, 0, world)
    tr.synchronize()
tr.close()
