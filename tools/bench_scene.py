#!/usr/bin/env python3
"""Time the other BASELINE configs on their stand-ins (not the bench.py headline) and report the demand
bytes SURVEY 8(d) item (ii) asks for:
    python tools/bench_scene.py 3 16          # config 3 stand-in: dragon x9 in the Cornell box, 1920x1080, 16 spp, 4 bounces
    BS_W=3840 BS_H=2160 BS_BOUNCES=8 python tools/bench_scene.py 11 64    # config 5 stand-in at full size
    python tools/bench_scene.py 0 8           # config 4 stand-in: 200 textured meshes
    BS_MESHES=340 BS_DETAIL=8 python tools/bench_scene.py 0 8    # the same at sponza.obj's size (261 k triangles)
    BS_HETERO=1 python tools/bench_scene.py 0 8    # the heterogeneous stand-in: sponza.mtl + 25 of its textures, 393 groups, 5 transforms
Options: BS_OPTS=name=value,... (rt_set_option, upload-time ones too), BS_FRAMES (timed frames, default: whole launches, >= 12),
BS_BATCH (frames per launch of rt_render_frames, default 4; 1 = one launch per frame), BS_JSON=path (also write the
figures as JSON), BS_COUNTERS=0 (skip the counter frame), BS_DEVICE_BUILD=1 (SAH searches on the GPU).
Demand bytes per segment = node tests x 48 + triangle tests x 96 + meshes x 240 (the reference's records: Node 48 B,
PackedTriangle 96 B, MeshUniform 240 B; wgsl:307,322 counters) -- what the shader's loop asks of memory per ray."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    g = os.path.join(ROOT, "tests", "golden")
    t0 = time.perf_counter()
    if n == 0 and os.environ.get("BS_HETERO"):   # config 4 stand-in with sponza.mtl's materials and textures (round 5)
        sc = scenes.sponza_hetero()
        name = "heterogeneous sponza stand-in (393 groups of 2 .. 40,000 triangles, 5 transforms, sponza.mtl + 25 of its textures)"
    elif n == 0:   # config 4 stand-in: many textured meshes
        nm, detail = int(os.environ.get("BS_MESHES", 200)), int(os.environ.get("BS_DETAIL", 1))
        sc = scenes.sponza_standin(nm, detail=detail)
        name = f"sponza stand-in ({nm} textured meshes, {12 * detail * detail} triangles each)"
    else:
        sc = scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                   scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n,
                                   device=0 if os.environ.get("BS_DEVICE_BUILD") else None)
        name = f"dragon.obj x{n * n} in the Cornell box"
    build_s = time.perf_counter() - t0
    arrays = rt.SceneArrays.from_scene(sc)
    W, H = (int(os.environ.get("BS_W", 1920)), int(os.environ.get("BS_H", 1080)))
    NB = int(os.environ.get("BS_BOUNCES", 4))
    batch = max(1, int(os.environ.get("BS_BATCH", 4)))
    frames = int(os.environ.get("BS_FRAMES", 0)) or batch * max(2, 12 // batch)   # whole launches of `batch` frames
    per_launch = -(-frames // -(-frames // batch))   # rt_render_frames cuts n frames into equal batches
    tr = rt.RayTracer(0, W, H)
    for kv in os.environ.get("BS_OPTS", "").split(","):   # e.g. BS_OPTS=forest=0,tlas=0 (upload-time options too)
        if kv:
            k, v = kv.split("=")
            tr.set_option(k, int(v))
    tr.set_option("batch_frames", batch)
    tr.load_scene(arrays)
    n_tri, n_nodes, n_mesh = arrays.triangles.shape[0], arrays.nodes.shape[0], arrays.meshes.shape[0]
    print(f"{name}: triangles {n_tri}, nodes {n_nodes}, meshes {n_mesh}; host build {build_s:.2f} s", flush=True)
    out = {"scene": name, "triangles": n_tri, "nodes": n_nodes, "meshes": n_mesh, "width": W, "height": H, "spp": spp,
           "bounces": NB, "frames_per_launch": per_launch}
    # warm-up (also gives the tile order), then timed frames
    tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=0), max(2, batch))
    ts = []
    for rep in range(3):
        tr.synchronize()
        tr.reset_timing()
        t1 = time.perf_counter()
        tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=2 + rep * frames), frames)
        tr.synchronize()
        ts.append((time.perf_counter() - t1) / frames * 1e3)
        st = tr.stats()
    out["frames_total"] = max(2, batch) + 3 * frames   # (every frame this run renders before the counter frame: tools/pmc_scene.sh)
    ms = statistics.median(ts)
    rays = st.segments / st.frames
    out.update(ms_per_frame=ms, kernel_ms_per_frame=st.kernel_ms / st.frames, rays_per_frame=rays,
               rays_traversed_per_frame=(st.segments - st.segments_reused) / st.frames, mrays_per_s=rays / ms / 1e3)
    print(f"{ms:.3f} ms/frame ({st.kernel_ms / st.frames:.3f} in kernels, {per_launch} frames per launch), {rays / ms / 1e3:.0f} Mrays/s, "
          f"{rays / 1e6:.2f} Mrays/frame ({(st.segments - st.segments_reused) / st.frames / 1e6:.2f} M traversed)", flush=True)
    if os.environ.get("BS_COUNTERS", "1") != "0":
        # one frame with the shader's stats counters (wgsl:307,322) compiled in
        tr.set_counters(True)
        tr.reset_timing()
        tr.render(rt.make_params(W, H, NB, spp, skybox=1, frames=1))
        sc_ = tr.stats()
        tr.set_counters(False)
        seg = max(sc_.segments, 1)
        nt, tt = sc_.node_tests / seg, sc_.triangle_tests / seg
        demand = nt * 48 + tt * 96 + n_mesh * 240
        out.update(node_tests_per_segment=nt, triangle_tests_per_segment=tt, demand_bytes_per_segment=demand,
                   demand_bytes_per_frame=demand * sc_.segments,
                   scene_bytes_reference_layout=n_nodes * 48 + n_tri * 96 + n_mesh * 240)
        print(f"per segment: {nt:.1f} box tests, {tt:.1f} triangle tests -> demand {demand:.0f} B/segment = "
              f"{demand * sc_.segments / 1e9:.2f} GB/frame ({demand * sc_.segments / (ms * 1e-3) / 1e12:.2f} TB/s at this frame time)", flush=True)
    if os.environ.get("BS_JSON"):
        json.dump(out, open(os.environ["BS_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
