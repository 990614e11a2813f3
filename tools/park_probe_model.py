#!/usr/bin/env python3
"""Offline model of a longer inline probe in front of a deferred walk (rt_kernel.hip: ITEM_DEFER): for a sample of the
rays parked in a round (tools/park_counts.py with PARK_SAMPLE=...), walk the deferred mesh's BVH on the CPU with the
closest distance at INF -- the walk's own tests until it reaches a leaf -- depth first, near child first, and count the
box-pair visits until (a) the ray has been shown to miss every leaf box (no park needed) or (b) it reaches a leaf
(park).  Prints what a budget of B visits would have saved."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

sample = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "park_sample.npz"))
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = os.path.join(ROOT, "tests", "golden")
arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                         scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=n_sub))
meshes = arrays.meshes
big = int(np.argmax(meshes["triangles"]))
m = meshes[big]
w2m = np.array(m["world_to_model"], np.float64)   # [col][row]
nd = arrays.nodes[int(m["node_offset"]):]
INF = float(2.0 ** 127)


def box(lo, inv, mn, mx):
    t1 = (mn - lo) * inv
    t2 = (mx - lo) * inv
    tn = np.max(np.minimum(t1, t2))
    tf = np.min(np.maximum(t1, t2))
    return tn if (tf >= tn and tn < INF and tf > 0) else INF


visits_miss, visits_hit = [], []
ro, rd = sample["ro"].astype(np.float64), sample["rd"].astype(np.float64)
K = min(len(ro), 6000)
for i in range(K):
    lo = w2m[0, :3] * ro[i, 0] + w2m[1, :3] * ro[i, 1] + w2m[2, :3] * ro[i, 2] + w2m[3, :3]
    ld = w2m[0, :3] * rd[i, 0] + w2m[1, :3] * rd[i, 1] + w2m[2, :3] * rd[i, 2]
    ld = ld / np.sqrt(np.dot(ld, ld))
    with np.errstate(divide="ignore"):
        inv = 1.0 / ld
    stack = [0]
    visits = 0
    reached = False
    while stack:
        k = stack.pop()
        node = nd[k]
        if node["count"] > 0:
            reached = True
            break
        visits += 1
        a, b = nd[node["left"]], nd[node["right"]]
        da, db = box(lo, inv, a["aabb_min"].astype(np.float64), a["aabb_max"].astype(np.float64)), box(lo, inv, b["aabb_min"].astype(np.float64), b["aabb_max"].astype(np.float64))
        near, far = (node["left"], node["right"]) if da < db else (node["right"], node["left"])
        dn, df = (da, db) if da < db else (db, da)
        if df < INF:
            stack.append(int(far))
        if dn < INF:
            stack.append(int(near))
    (visits_hit if reached else visits_miss).append(visits)
vm, vh = np.array(visits_miss), np.array(visits_hit)
print(f"{K} parked rays: {len(vm)} never reach a leaf box ({len(vm) / K:.3f}), {len(vh)} reach one ({len(vh) / K:.3f}); "
      f"GPU walk results of the sample: {sample['hit'][:K].mean():.3f} hit a triangle")
print(f"visits until the probe knows: miss rays median {np.median(vm) if len(vm) else 0:.0f}, p90 {np.percentile(vm, 90) if len(vm) else 0:.0f}, max {vm.max() if len(vm) else 0}; "
      f"rays that reach a leaf: median {np.median(vh):.0f}, p90 {np.percentile(vh, 90):.0f}")
for B in (3, 4, 6, 8, 12, 16, 24, 32):
    saved = (vm <= B).sum()
    cost = np.minimum(vm, B).sum() + np.minimum(vh, B).sum()
    print(f"budget {B:2d} visits: {saved / K:.3f} of the parks avoided; {cost / K:.1f} inline visits per probed ray")

# how many parks a cut of the root box against the closest hit so far would spare (the small meshes come first in the item
# order): the ray enters the deferred mesh's root box beyond that hit.  World distance = local distance here (the stand-in's
# dragon has a similarity transform of scale s: t_world = t_local * s).
if "closest" in sample.files:
    root = nd[0]
    c0 = np.array(m["model_to_world"], np.float64)[0, :3]
    scale = float(np.sqrt(np.dot(c0, c0)))
    beyond = 0
    for i in range(K):
        lo = w2m[0, :3] * ro[i, 0] + w2m[1, :3] * ro[i, 1] + w2m[2, :3] * ro[i, 2] + w2m[3, :3]
        ld = w2m[0, :3] * rd[i, 0] + w2m[1, :3] * rd[i, 1] + w2m[2, :3] * rd[i, 2]
        ld = ld / np.sqrt(np.dot(ld, ld))
        with np.errstate(divide="ignore"):
            inv = 1.0 / ld
        entry = box(lo, inv, root["aabb_min"].astype(np.float64), root["aabb_max"].astype(np.float64))
        if entry < INF and entry * scale * 1.0 > float(sample["closest"][i]) * 1.125:
            beyond += 1
    print(f"root box entered beyond 1.125 x the closest hit so far: {beyond / K:.3f} of the parked rays "
          f"(closest known at park time: {np.isfinite(sample['closest'][:K]).mean():.3f} finite, median {np.median(sample['closest'][:K]):.3g})")
