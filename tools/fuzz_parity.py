#!/usr/bin/env python3
"""Stress version of tests/test_gpu_scenes.py::test_random_scenes: N seeded random scenes, GPU (both kernel
variants, with and without counters, one launch per frame; and three accumulating frames as one batch of
rt_render_frames; and both again with the scene's biggest BVH mesh deferred to rt_walk_kernel, with and without the
hybrid small blob; the general kernels where the specialised ones are the default) against the CPU oracle, bit for bit.
Every fourth scene is a many-mesh one (top-level trees; cross-mesh pruning on, which is the default, and off; in the
experiments build also through the wavefront sequence).  FUZZ_MANY=1: every scene is a many-mesh one.  FUZZ_SHEAR=1: the
meshes' matrices are post-multiplied by a seeded shear and non-uniform scale (so that model_to_world is no longer a
similarity, and only approximately the inverse of world_to_model in float32 -- the cases cross-mesh pruning's bound on
the world distance has to hold for).  usage: fuzz_parity.py [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import ray_tracer_2_amd as rt  # noqa: E402
from oracle import oracle  # noqa: E402
from test_gpu_scenes import _random_scene  # noqa: E402

EXPERIMENTS = b"+experiments" in rt.load().rt_version()
MANY, SHEAR = bool(int(os.environ.get("FUZZ_MANY", 0))), bool(int(os.environ.get("FUZZ_SHEAR", 0)))


def sheared(arrays, seed):
    """The scene with every run of meshes that share a local space moved into a sheared, non-uniformly scaled one:
    model_to_world' = model_to_world * A, world_to_model' = inv(A) * world_to_model (float64, rounded to float32)."""
    rng = np.random.default_rng(seed)
    m = arrays.meshes.copy()
    done = {}
    for i in range(m.shape[0]):
        key = m["world_to_model"][i].tobytes()
        if key not in done:
            A = np.eye(4)
            A[:3, :3] = np.diag(rng.uniform(0.4, 2.5, 3)) @ (np.eye(3) + np.triu(rng.uniform(-0.6, 0.6, (3, 3)), 1))
            # (stored [col][row]: the array is the transpose of the matrix)
            w2m = np.linalg.inv(A) @ arrays.meshes["world_to_model"][i].astype(np.float64).T
            m2w = arrays.meshes["model_to_world"][i].astype(np.float64).T @ A
            done[key] = (w2m.T.astype(np.float32), m2w.T.astype(np.float32))
        m["world_to_model"][i], m["model_to_world"][i] = done[key]
    return rt.SceneArrays(arrays.uniform, arrays.spheres, m, arrays.triangles, arrays.nodes, arrays.textures)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tr = rt.RayTracer(0, 256 * int(os.environ.get("FUZZ_SCALE", 1)), 256 * int(os.environ.get("FUZZ_SCALE", 1)))
bad = 0
for seed in range(first, first + count):
    arrays = _random_scene(rt, seed, many=MANY or seed % 4 == 3)   # every fourth: 5-40 meshes per transform group (top-level trees)
    if SHEAR:
        arrays = sheared(arrays, seed)
    tr.set_option("tlas_min", 2 if seed % 8 == 7 else 8)
    scale = int(os.environ.get("FUZZ_SCALE", 1))   # larger frames: many tiles per resident wave, refill and pipelining at work
    w, h = scale * (64 + 8 * (seed % 9)) + (seed % 5 if scale > 1 else 0), scale * (40 + 4 * (seed % 7)) + (seed % 3 if scale > 1 else 0)
    p = rt.make_params(w, h, 1 + seed % 6, 1 + seed % 4, skybox=seed % 2, frames=0)
    ref, st = oracle.render(p, arrays)
    tr.load_scene(arrays)
    for variant in (0, 1):
        tr.set_option("kernel_variant", variant)
        for counters in (True, False):
            tr.set_counters(counters)
            tr.reset_timing()
            tr.render(p)
            gpu = tr.read_image(w, h)
            s = tr.stats()
            ok = np.array_equal(gpu.view(np.uint32), ref.view(np.uint32)) and s.segments == st.segments
            if counters:
                ok = ok and (s.node_tests, s.triangle_tests) == (st.node_tests, st.triangle_tests)
            if not ok:
                bad += 1
                print(f"MISMATCH seed {seed} variant {variant} counters {counters}: {int((gpu.view(np.uint32) != ref.view(np.uint32)).sum())} words differ")
    tr.set_counters(False)
    tr.set_option("kernel_variant", -1)
    # three accumulating frames: sequentially on the oracle, as one overlapped batch on the GPU
    acc = np.zeros((h, w, 4), np.float32)
    for f in range(3):
        p.frames = f
        acc, _ = oracle.render(p, arrays, image=acc)
    p.frames = 0
    tr.write_image(np.zeros((h, w, 4), np.float32))
    tr.render_frames(p, 3)
    if not np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32)):
        bad += 1
        print(f"MISMATCH seed {seed} rt_render_frames")
    # (round 5) the scene read from global memory, whatever its size: the primary-ray memo read in place from the primary table
    # (memo_in_table = 1, the default) and copied into the per-wave buffer (0); one batch and three single-frame calls each
    tr.set_option("lds_scene", 0)
    for mit in (1, 0):
        tr.set_option("memo_in_table", mit)
        tr.write_image(np.zeros((h, w, 4), np.float32))
        tr.render_frames(p, 3)
        ok = np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32))
        tr.write_image(np.zeros((h, w, 4), np.float32))
        for f in range(3):
            p.frames = f
            tr.render(p)
        p.frames = 0
        if not (ok and np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32))):
            bad += 1
            print(f"MISMATCH seed {seed} global memory, memo_in_table = {mit}")
    tr.set_option("memo_in_table", 1)
    tr.set_option("lds_scene", 1)
    # deferred walks: the biggest BVH mesh of the scene, however small, walked by rt_walk_kernel (few-mesh scenes)
    if not (MANY or seed % 4 == 3):
        tr.set_option("defer_min_nodes", 1)
        tr.set_option("sort_rounds", 1 + seed % 5)
        tr.load_scene(arrays)
        tr.set_counters(True)
        tr.reset_timing()
        tr.write_image(np.zeros((h, w, 4), np.float32))
        p.frames = 0
        tr.render(p)
        s = tr.stats()
        if not (np.array_equal(tr.read_image(w, h).view(np.uint32), ref.view(np.uint32)) and
                (s.segments, s.node_tests, s.triangle_tests) == (st.segments, st.node_tests, st.triangle_tests)):
            bad += 1
            print(f"MISMATCH seed {seed} deferred walks (single frame, counters)")
        tr.set_counters(False)
        for hybrid in ((0, 1) if EXPERIMENTS else (0,)):   # (experiments build) the parking launches on the LDS-staged small blob
            tr.set_option("hybrid", hybrid)
            tr.write_image(np.zeros((h, w, 4), np.float32))
            tr.render_frames(p, 3)
            if not np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32)):
                bad += 1
                print(f"MISMATCH seed {seed} deferred walks (batch, hybrid {hybrid})")
        tr.set_option("hybrid", 0)
        tr.set_option("sort_rounds", -1)
        tr.set_option("defer_min_nodes", 1024)
        # (round 3) the general kernels where the specialised instantiation is the default
        tr.set_option("specialise", 0)
        tr.load_scene(arrays)
        tr.write_image(np.zeros((h, w, 4), np.float32))
        tr.render_frames(p, 3)
        if not np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32)):
            bad += 1
            print(f"MISMATCH seed {seed} specialise = 0")
        tr.set_option("specialise", 1)
    else:
        # (round 3) many-mesh scenes through the wavefront sequence: single frame with counters, and the batch
        if EXPERIMENTS:
            tr.set_option("wavefront", 1)
            tr.set_counters(True)
            tr.reset_timing()
            p.frames = 0
            tr.render(p)
            s = tr.stats()
            used = tr.last_launch()["wavefront"]
            if used and not (np.array_equal(tr.read_image(w, h).view(np.uint32), ref.view(np.uint32)) and
                             (s.segments, s.node_tests, s.triangle_tests) == (st.segments, st.node_tests, st.triangle_tests)):
                bad += 1
                print(f"MISMATCH seed {seed} wavefront (single frame, counters)")
            tr.set_counters(False)
            tr.write_image(np.zeros((h, w, 4), np.float32))
            tr.render_frames(p, 3)
            if not np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32)):
                bad += 1
                print(f"MISMATCH seed {seed} wavefront (batch)")
            tr.set_option("wavefront", 0)
        # (round 4) cross-mesh pruning off: the many-mesh product kernels walking every mesh from an infinite distance
        tr.set_option("cross_prune", 0)
        tr.write_image(np.zeros((h, w, 4), np.float32))
        tr.render_frames(p, 3)
        if not np.array_equal(tr.read_image(w, h).view(np.uint32), acc.view(np.uint32)):
            bad += 1
            print(f"MISMATCH seed {seed} cross_prune = 0")
        tr.set_option("cross_prune", int(os.environ.get("FUZZ_PRUNE", "1")))
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} scenes, {bad} mismatches", flush=True)
print(f"{count} scenes, {bad} mismatches")
sys.exit(1 if bad else 0)
