#!/usr/bin/env python3
"""Frame times of every BASELINE config on its stand-in, at the frames-per-launch settings a host would use, written as the
tracked table profiles/<tag>_standins.txt (wall time of the whole sequence of launches / frames; default options).

    python tools/standins_table.py r03            # on the GPU box; writes gpurun_out/r03_standins.txt (copy to profiles/)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import scenes  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def dragon(n):
    return rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(G, "cornell_raw.npz")),
                                                           scenes.load_raw_meshes(os.path.join(G, "dragon_raw.npz")),
                                                           subdivide=n, device=0 if n > 3 else None))


def measure(arrays, w, h, spp, bounces, batch, frames, reps=3, opts=()):
    tr = rt.RayTracer(0, w, h)
    for k, v in opts:
        tr.set_option(k, v)
    tr.set_option("batch_frames", max(1, batch))
    tr.load_scene(arrays)
    p = rt.make_params(w, h, bounces, spp, skybox=1, frames=0)

    def run(f0, n):
        p.frames = f0
        if batch > 1:
            tr.render_frames(p, n)
        else:
            for f in range(n):
                p.frames = f0 + f
                tr.render(p)
    run(0, max(batch, 2))
    ts = []
    for r in range(reps):
        tr.synchronize()
        tr.reset_timing()
        t0 = time.perf_counter()
        run(max(batch, 2) + r * frames, frames)
        tr.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3 / frames)
        st = tr.stats()
    info = tr.last_launch()
    rays, reused = st.segments / st.frames, st.segments_reused / st.frames
    tr.close()
    return statistics.median(ts), rays, reused, info


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out = []
    say = lambda s: (out.append(s), print(s, flush=True))
    say(f"Stand-in frame times, {tag} (tools/standins_table.py; one MI355X; wall time per frame of back-to-back launches, rt_synchronize at the end,")
    say("median of 3 repetitions; default options; [n] = frames per launch, [1] = one rt_render per frame, consecutive frames pipelined).  The named assets of")
    say("BASELINE configs 3-5 are absent from the reference checkout (.MISSING_LARGE_BLOBS): these are the stand-ins of SURVEY.md 8(d).")
    say("")
    PR = (("cross_prune", 1),)   # the opt-in pruned many-mesh kernels, listed beside the default
    cases = [
        ("config 2: CornellBox-Original, 1920x1080, 8 spp, 4 bounces", lambda: rt.SceneArrays.load(os.path.join(G, "cornell_scene.npz")),
         1920, 1080, 8, 4, [(1, 32), (8, 32), (32, 64), (64, 128)]),
        ("config 3 stand-in: dragon.obj x9 (78,408 triangles) in the Cornell box, 1920x1080, 16 spp, 4 bounces", lambda: dragon(3),
         1920, 1080, 16, 4, [(1, 8), (8, 16), (32, 32), (64, 64)]),
        ("config 4 stand-in: 200 textured meshes x 12 triangles + quad + sphere, 1920x1080, 8 spp, 4 bounces",
         lambda: rt.SceneArrays.from_scene(scenes.sponza_standin(200)), 1920, 1080, 8, 4, [(1, 8), (8, 16), (8, 16, PR)]),
        ("config 4 stand-in at sponza.obj's size: 340 meshes x 768 triangles (261 k triangles), 1920x1080, 8 spp, 4 bounces",
         lambda: rt.SceneArrays.from_scene(scenes.sponza_standin(340, detail=8)), 1920, 1080, 8, 4, [(1, 8), (8, 16), (64, 64), (1, 8, PR), (64, 64, PR)]),
        ("config 4 stand-in, heterogeneous (round 5): sponza.mtl + 25 of its textures, 393 groups of 2 .. 40,000 triangles (259 k), 5 transforms, 1920x1080, 8 spp, 4 bounces",
         lambda: rt.SceneArrays.from_scene(scenes.sponza_hetero()), 1920, 1080, 8, 4, [(1, 8), (8, 16), (64, 64), (64, 64, PR)]),
        ("config 5 geometry: dragon.obj x121 (1,054,152 triangles), 1920x1080, 8 spp, 4 bounces", lambda: dragon(11),
         1920, 1080, 8, 4, [(1, 8), (16, 32), (64, 64)]),
        ("config 5 stand-in at its BASELINE size: dragon.obj x121, 3840x2160, 64 spp, 8 bounces", None,
         3840, 2160, 64, 8, [(1, 2), (4, 4), (8, 8), (32, 32)]),
    ]
    last = None
    for name, make, w, h, spp, nb, runs in cases:
        arrays = make() if make else last
        last = arrays
        say(name)
        for run_ in runs:
            batch, frames, opts = (*run_, ())[:3]
            ms, rays, reused, info = measure(arrays, w, h, spp, nb, batch, frames, opts=opts)
            kind = "deferred walks" if info["deferred_walks"] else "wavefront" if info["wavefront"] else "many-mesh kernels" if info["many_mesh"] else "few-mesh kernels"
            say(f"  [{batch:2d}] {ms:9.3f} ms/frame   {rays / ms / 1e3:8.0f} Mrays/s ({rays / 1e6:.2f} M rays per frame, {(rays - reused) / 1e6:.2f} M traversed)   "
                f"{kind}{', specialised' if info['specialised'] else ''}{', scene in LDS' if info['scene_in_lds'] else ''}"
                f"{'   [' + ', '.join(f'{k} = {v}' for k, v in opts) + ']' if opts else ''}   {info['device_mb_held']} MiB held")
        say("")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_standins.txt"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
