#!/usr/bin/env python3
"""Time other BASELINE configs (not the bench.py headline): config 3 stand-in
(dragon x9 in the Cornell box, 1920x1080, 16 spp, 4 bounces)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    g = os.path.join(ROOT, "tests", "golden")
    if n == 0:   # config 4 stand-in: many textured meshes
        sc = scenes.sponza_standin(200)
    else:
        sc = scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                   scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n)
    arrays = rt.SceneArrays.from_scene(sc)
    W, H = (int(os.environ.get("BS_W", 1920)), int(os.environ.get("BS_H", 1080)))   # BS_W=3840 BS_H=2160 BS_BOUNCES=8: config 5 at full size
    NB = int(os.environ.get("BS_BOUNCES", 4))
    tr = rt.RayTracer(0, W, H)
    for kv in os.environ.get("BS_OPTS", "").split(","):   # e.g. BS_OPTS=forest=0,tlas=0 (upload-time options too)
        if kv:
            k, v = kv.split("=")
            tr.set_option(k, int(v))
    tr.load_scene(arrays)
    print(f"triangles {arrays.triangles.shape[0]}, nodes {arrays.nodes.shape[0]}, meshes {arrays.meshes.shape[0]}")
    for variant in ((0,) if os.environ.get("BS_W") else (0, 1)):
        tr.set_option("kernel_variant", variant)
        ts, rays = [], 0
        for r in range(4):
            tr.reset_timing()
            for f in range(3):
                tr.render(rt.make_params(W, H, NB, spp, skybox=1, frames=1 + f))
            st = tr.stats()
            if r:
                ts.append(st.kernel_ms / st.launches)
                rays = st.segments / st.launches
        t = statistics.median(ts)
        print(f"variant {variant}: {t:.3f} ms/frame, {rays / t / 1e3:.0f} Mrays/s, {rays / 1e6:.1f} Mrays/frame")


if __name__ == "__main__":
    main()
