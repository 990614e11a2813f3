#!/usr/bin/env python3
"""Pass census of the wavefront walk kernel (diagnostic build, -DRT_DIAG=1): how many passes of each kind a frame takes
and how many lanes each serves.   python tools/wf_diag.py [scene=sponza340] [w h]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_2_amd import build
DIAG_SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x_diag.so")
if "--build-only" in sys.argv:
    build.build_product(extra_flags=("-DRT_DIAG=1",), out=DIAG_SO)
    sys.exit(0)
import ray_tracer_2_amd.lib as lib
lib.LIB_PATH = DIAG_SO
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "sponza340"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
n = int(scene[6:])
arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(n, detail=8 if n >= 300 else 1))
tr = rt.RayTracer(0, W, H)
tr.set_option("wavefront", 1)
tr.load_scene(arrays)
L = rt.load()
buf = (C.c_uint64 * 128)()
p = rt.make_params(W, H, 4, 8, skybox=1, frames=0)
tr.render(p)
L.rt_diag_read(tr._h, buf, 1)
tr.reset_timing()
p.frames = 1
tr.render(p)
st = tr.stats()
L.rt_diag_read(tr._h, buf, 1)
names = {20: "BOX pass", 27: "  of which tree-node lanes", 21: "LEAF pass", 22: "ADVANCE pass", 24: "  offers", 25: "  items", 26: "  results stored",
         23: "refill pass"}
rays = buf[2 * 26 + 1]
print(f"{scene} {W}x{H} 8 spp 4 bounces: segments {st.segments}, reused {st.segments_reused}, rays walked {rays}, kernel {st.kernel_ms:.2f} ms")
tot = sum(buf[2 * k] for k in (20, 21, 22, 23))
for k, name in names.items():
    v, l = buf[2 * k], buf[2 * k + 1]
    if v:
        print(f"{name:28s} passes {v:11d} ({v / tot:6.1%})  lanes per pass {l / v:5.1f}  lane-steps per ray {l / max(rays, 1):6.2f}")
