#!/usr/bin/env python3
"""A/B timing of two (or more) BUILDS of the product library on one scene: each build runs in its own child process
(one library per process; the parent never touches the GPU), the builds are interleaved over several rounds, and the
per-frame time is taken from the HIP events around the launches (rt_get_stats).

    python tools/ab_lib.py [--scene cornell|dragon3|dragon11|sponza200|sponza340] [--spp 8] [--bounces 4] [--batch 32]
                           [--frames 64] [--rounds 3] [--w 1920 --h 1080] [--opts name=v,...] lib_a.so lib_b.so ...

A library path may carry its own options: `path.so:name=v,name=v` (rt_set_option, before the scene upload).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import ray_tracer_2_amd as rt
    from ray_tracer_2_amd import scenes
    g = os.path.join(ROOT, "tests", "golden")
    if a.scene == "cornell":
        arrays = rt.SceneArrays.load(os.path.join(g, "cornell_scene.npz"))
    elif a.scene.startswith("dragon"):
        n = int(a.scene[6:])
        arrays = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                                 scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")),
                                                                 subdivide=n, device=0 if n > 3 else None))
    else:
        n = int(a.scene[6:])
        arrays = rt.SceneArrays.from_scene(scenes.sponza_standin(n, detail=a.detail if a.detail else (8 if n >= 300 else 1)))
    tr = rt.RayTracer(0, a.w, a.h)
    for kv in (a.opts or "").split(","):
        if kv:
            k, v = kv.split("=")
            tr.set_option(k, int(v))
    tr.set_option("batch_frames", max(1, a.batch))
    tr.load_scene(arrays)
    p = rt.make_params(a.w, a.h, a.bounces, a.spp, skybox=1, frames=0)

    def run(f0, n):
        p.frames = f0
        if a.batch > 1:
            tr.render_frames(p, n)
        else:
            for f in range(n):
                p.frames = f0 + f
                tr.render(p)
    run(0, max(a.batch, 8))     # warm-up (tile order, tables)
    tr.synchronize()
    tr.reset_timing()
    import time
    t0 = time.perf_counter()
    run(max(a.batch, 8), a.frames)
    tr.synchronize()
    wall = (time.perf_counter() - t0) / a.frames * 1e3
    st = tr.stats()
    print(json.dumps({"ms_per_frame": st.kernel_ms / st.frames, "wall_ms_per_frame": wall, "segments_per_frame": st.segments / st.frames,
                      "launches": st.launches, "launch": tr.last_launch() if hasattr(tr, "last_launch") else None}), flush=True)
    tr.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--opts", default="")
    ap.add_argument("--detail", type=int, default=0, help="sponza stand-in: triangles per mesh = 12 x detail^2 (default 8 from 300 meshes, else 1)")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    libs = a.libs or [os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x.so")]
    times = {l: [] for l in libs}
    segs, info, walls = {}, {}, {}
    for r in range(a.rounds):
        for l in libs:
            path, _, own = l.partition(":")
            env = dict(os.environ, RT2_LIB=os.path.abspath(path))
            opts = ",".join(x for x in (a.opts, own) if x)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--scene", a.scene, "--spp", str(a.spp), "--bounces", str(a.bounces),
                   "--batch", str(a.batch), "--frames", str(a.frames), "--w", str(a.w), "--h", str(a.h), "--detail", str(a.detail), "--opts", opts]
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(l, "FAILED:", out.stderr[-400:], flush=True)
                continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            times[l].append(d["ms_per_frame"])
            walls.setdefault(l, []).append(d.get("wall_ms_per_frame", float("nan")))
            segs[l] = d["segments_per_frame"]
            info[l] = d.get("launch")
    print(f"scene {a.scene} {a.w}x{a.h} {a.spp} spp {a.bounces} bounces, {a.batch} frames per launch, {a.frames} frames x {a.rounds} rounds")
    for l in libs:
        if times[l]:
            print(f"{os.path.basename(l):60s} median {statistics.median(times[l]):8.4f}  min {min(times[l]):8.4f} ms/frame (sum of launch times; wall {statistics.median(walls[l]):.4f})   rays/frame {segs[l]:.0f}  {info.get(l)}", flush=True)


if __name__ == "__main__":
    main()
