#!/usr/bin/env python3
"""Site-by-site account of what the lanes of the render / walk kernels ask of global memory (round 5).

    python tools/mem_sites.py 0 8         # config 4 stand-in (BS_MESHES / BS_DETAIL as in tools/bench_scene.py)
    python tools/mem_sites.py 3 16        # config 3 stand-in;  BS_W / BS_H / BS_BOUNCES / BS_BATCH / BS_OPTS as there
    MS_PMC=gpurun_out/pmcs_<tag>/v1 python tools/mem_sites.py ...   # print the counters of tools/pmc_scene.sh beside it

Uses the census build (-DRT_DIAG=1, tools/diag.py): every global-memory access site of the kernels carries a DIAG(id) that
counts wave visits and active lanes; bytes = active lanes x the bytes one lane moves at that site (listed below, from the
source).  These are DEMAND bytes at the lanes (what L1 / L2 / Infinity Cache / HBM together have to serve); the PMC counters
(FETCH_SIZE x 2, WRITE_SIZE) are what crosses the L2's memory side.  Wave-uniform reads (the mesh loop's items and mesh
records, root-leaf triangles) are scalar loads through the constant cache and are not listed.
"""
import ctypes as C
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diag  # noqa: E402

# id -> (site, kind, bytes per active lane, source)
SITES = [
    (15, "pixel taken (refill)", "-", 0, "rt_render_persistent_kernel: pixel_begin"),
    (25, "memo copy: table entry read", "R", 64, "memo_from_table (pixel_cache = 2 only)"),
    (25, "memo copy: per-wave memo written", "W", 52, "memo_from_table (pixel_cache = 2 only)"),
    (20, "memo read in place from the table", "R", 16, "with_memo_ro (pixel_cache = 4): flags + ray per sample"),
    (21, "memo hit read (table entry / memo buffer)", "R", 40, "memo_hit_load"),
    (17, "top-level tree record", "R", 64, "load_tlas"),
    (7, "BVH wide record (render kernels)", "R", 64, "load_wide in traverse_mesh / forest / many-mesh walk"),
    (8, "leaf triangle record (render kernels)", "R", 48, "tri_test arguments"),
    (10, "model_to_world of a mesh hit", "R", 64, "world_hit"),
    (11, "winner: shading record + matrix", "R", 128, "isect_finish"),
    (13, "material of a diffuse hit", "R", 72, "path_end"),
    (14, "material of a glass hit", "R", 40, "path_end"),
    (26, "texture descriptor + 4 texels", "R", 32, "sample_texture"),
    (16, "image texel store", "W", 16, "pixel_finish / store_texel"),
    (22, "park record store", "W", 144, "park_store (9 planes)"),
    (23, "park record load + hit planes", "R", 176, "park_load + park_load_hit (11 planes)"),
    (43, "walk kernel: ray of a record", "R", 32, "rt_walk_kernel"),
    (43, "walk kernel: result store", "W", 16, "rt_walk_kernel (plane 13)"),
    (42, "walk kernel: BVH wide record", "R", 64, "rt_walk_kernel"),
    (40, "walk kernel: leaf triangle record", "R", 48, "rt_walk_kernel"),
]


def pmc_totals(prefix):
    """MB per kernel from one variant of tools/pmc_scene.sh (prefix = gpurun_out/pmcs_<tag>/v<k>)."""
    out = {}
    for ctr, k in (("FETCH_SIZE", 2048), ("WRITE_SIZE", 1024)):
        tot = {}
        for f in glob.glob(f"{prefix}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                tot[name] = tot.get(name, 0.0) + float(r["Counter_Value"]) * k / 1e6
        out[ctr] = tot
    return out


def main():
    diag.build_diag()
    import ray_tracer_2_amd.lib as lib
    lib.LIB_PATH = diag.DIAG_SO
    import ray_tracer_2_amd as rt
    from ray_tracer_2_amd import scenes
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    g = os.path.join(ROOT, "tests", "golden")
    if n == 0 and os.environ.get("BS_HETERO"):
        sc = scenes.sponza_hetero()
        name = "heterogeneous sponza stand-in (393 groups, 259 k triangles, 5 transforms)"
    elif n == 0:
        nm, detail = int(os.environ.get("BS_MESHES", 340)), int(os.environ.get("BS_DETAIL", 8))
        sc = scenes.sponza_standin(nm, detail=detail)
        name = f"sponza stand-in ({nm} meshes x {12 * detail * detail} triangles)"
    else:
        sc = scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                   scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n,
                                   device=0 if os.environ.get("BS_DEVICE_BUILD") else None)
        name = f"dragon.obj x{n * n} in the Cornell box"
    arrays = rt.SceneArrays.from_scene(sc)
    W, H = int(os.environ.get("BS_W", 1920)), int(os.environ.get("BS_H", 1080))
    NB = int(os.environ.get("BS_BOUNCES", 4))
    batch = max(1, int(os.environ.get("BS_BATCH", 8)))
    tr = rt.RayTracer(0, W, H)
    for kv in os.environ.get("BS_OPTS", "").split(","):
        if kv:
            tr.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    tr.set_option("batch_frames", batch)
    tr.load_scene(arrays)
    L = rt.load()
    buf = (C.c_uint64 * 128)()
    tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=0), batch)   # warm-up: tables, tile order
    tr.synchronize()
    L.rt_diag_read(tr._h, buf, 1)
    tr.reset_timing()
    tr.render_frames(rt.make_params(W, H, NB, spp, skybox=1, frames=batch), batch)
    L.rt_diag_read(tr._h, buf, 1)
    st = tr.stats()
    frames = batch
    print(f"{name}: {W}x{H}, {spp} spp, {NB} bounces, {batch} frames per launch; options [{os.environ.get('BS_OPTS', '')}]")
    print(f"launch: {tr.last_launch()}")
    print(f"per frame: {st.segments / frames / 1e6:.2f} M rays ({(st.segments - st.segments_reused) / frames / 1e6:.2f} M traversed), "
          f"{W * H / 1e6:.2f} M pixels, {W * H * spp / 1e6:.2f} M samples")
    print(f"{'site':46s} {'R/W':3s} {'M lanes/frame':>13s} {'avg lanes':>9s} {'B/lane':>6s} {'MB/frame':>10s}   source")
    tot = {"R": 0.0, "W": 0.0}
    rows = []
    for sid, site, kind, nbytes, src in SITES:
        visits, lanes = buf[2 * sid], buf[2 * sid + 1]
        if sid == 20:   # (memo_hit_load's reads are listed under 21: what is left are the flag + ray reads)
            lanes_eff = lanes
        else:
            lanes_eff = lanes
        if not visits:
            continue
        mb = lanes_eff * nbytes / frames / 1e6
        if kind in tot:
            tot[kind] += mb
        rows.append((site, kind, lanes_eff / frames / 1e6, lanes / visits, nbytes, mb, src))
        print(f"{site:46s} {kind:3s} {lanes_eff / frames / 1e6:13.3f} {lanes / visits:9.1f} {nbytes:6d} {mb:10.1f}   {src}")
    print(f"demand at the lanes per frame: read {tot['R']:.1f} MB, write {tot['W']:.1f} MB (vector memory instructions only)")
    out = {"scene": name, "width": W, "height": H, "spp": spp, "bounces": NB, "frames_per_launch": batch,
           "options": os.environ.get("BS_OPTS", ""), "sites": rows, "demand_read_mb": tot["R"], "demand_write_mb": tot["W"]}
    pre = os.environ.get("MS_PMC")
    if pre:
        p = pmc_totals(pre)
        try:
            j = json.load(open(pre + "_WRITE_SIZE.json"))
            fr = None
        except Exception:
            j = {}
        print("PMC (tools/pmc_scene.sh, all launches of its run; FETCH_SIZE x 2):")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            for kname, mb in sorted(p[ctr].items(), key=lambda kv: -kv[1])[:6]:
                print(f"    {ctr:10s} {kname[:80]:80s} {mb:12.1f} MB")
        out["pmc"] = p
    if os.environ.get("MS_JSON"):
        json.dump(out, open(os.environ["MS_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
