#!/usr/bin/env python3
"""Divergence census (diagnostic build, -DRT_DIAG=1): for each code section,
how many times a wave entered it and how many lanes were active on average.
    python tools/diag.py [variant]
Builds ray_tracer_2_amd/librt2_mi355x_diag.so on first use (hipcc).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ray_tracer_2_amd import build  # noqa: E402

DIAG_SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x_diag.so")
SECTIONS = {0: "iteration (path_step)", 1: "sample start (ray gen)", 3: "mesh transform",
            4: "root-leaf tri: det", 5: "root-leaf tri: body", 6: "bvh: node visit", 7: "bvh: internal (2 aabb)",
            8: "bvh-leaf tri: det", 9: "bvh-leaf tri: body", 10: "mesh hit -> world", 11: "winner finalize",
            12: "miss: sky", 13: "shade: diffuse", 14: "shade: glass", 15: "refill: pixel_begin",
            16: "pixel_finish", 28: "vote: nobody wants", 29: "vote: run, all want", 30: "vote: run, some reuse", 31: "vote: wait", 17: "top tree: node (2 aabb)", 18: "top tree: mesh entry", 19: "shared walk: loop trip"}


TIME_SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x_diagt.so")
TIMED = {0: "intersect_scene (all)", 1: "sample start (ray gen / memo ray)", 2: "spheres", 3: "mesh transform",
         4: "root-leaf meshes", 5: "forest: root-box marks", 6: "forest: node visits", 7: "forest: leaf triangles",
         8: "forest: deal tasks, fetch rays", 9: "winner finalize", 10: "miss: sky", 11: "memo hit load",
         12: "shade (hit)", 13: "memo hit store", 14: "refill / tile pull", 15: "path_step (all)",
         16: "single-mesh BVH walk", 17: "forest: world hit + results back", 18: "two-leaf meshes", 19: "mesh hit -> world, offer",
         20: "park: slot + record store", 21: "resume: record load (in refill)", 22: "resume: hit load, offer, finish"}


def build_timed():
    build.build_product(extra_flags=("-DRT_DIAGT=1",), out=TIME_SO)


def build_diag():
    build.build_product(extra_flags=("-DRT_DIAG=1",), out=DIAG_SO)


def main():
    timed = "--time" in sys.argv
    if timed:
        sys.argv.remove("--time")
        build_timed()
    else:
        build_diag()
    if "--build-only" in sys.argv:
        return
    import ray_tracer_2_amd.lib as lib
    lib.LIB_PATH = TIME_SO if timed else DIAG_SO
    import ray_tracer_2_amd as rt
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (960, 540)
    if os.environ.get("DIAG_SCENE") == "dragon":
        from ray_tracer_2_amd import scenes
        g = os.path.join(ROOT, "tests", "golden")
        arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                                 scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), 3))
    elif os.environ.get("DIAG_SCENE") == "sponza":
        from ray_tracer_2_amd import scenes
        arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(int(os.environ.get("BS_MESHES", 200)), detail=int(os.environ.get("BS_DETAIL", 1))))
    else:
        arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
    tr = rt.RayTracer(0, W, H)
    spp, batch = int(os.environ.get("DIAG_SPP", 8)), int(os.environ.get("DIAG_BATCH", 1))
    for kv in os.environ.get("DIAG_OPTS", "").split(","):
        if kv:
            tr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if batch > 1:
        tr.set_option("batch_frames", batch)
    tr.load_scene(arrays)
    tr.set_option("kernel_variant", variant)
    L = rt.load()
    buf = (C.c_uint64 * 128)()
    L.rt_diag_read(tr._h, buf, 1)
    if batch > 1:
        tr.render_frames(rt.make_params(W, H, 4, spp, skybox=1, frames=0), batch)
    else:
        tr.render(rt.make_params(W, H, 4, spp, skybox=1, frames=0))
    L.rt_diag_read(tr._h, buf, 1)
    print(tr.last_launch())
    if timed:
        tot = buf[40 + 15] + buf[40 + 14]
        print(f"variant {variant}: {W}x{H}, {spp} spp x {batch} frames, 4 bounces -- wave-cycles per section (s_memtime), share of path_step + refill")
        for k, name in TIMED.items():
            print(f"{name:36s} {buf[40 + k]:16d} {buf[40 + k] / tot:7.1%}")
        return
    it = buf[0]
    print(f"variant {variant}: {W}x{H}, 8 spp, 4 bounces")
    print(f"{'section':28s} {'wave visits':>12s} {'per iter':>9s} {'avg lanes':>9s} {'util':>6s}")
    for k, name in SECTIONS.items():
        v, l = buf[2 * k], buf[2 * k + 1]
        if v:
            print(f"{name:28s} {v:12d} {v / it:9.2f} {l / v:9.1f} {l / v / 64:6.1%}")


if __name__ == "__main__":
    main()
