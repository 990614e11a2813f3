#!/usr/bin/env python3
"""Balance of the multi-GPU strip split, measured on ONE GPU: every rank's share of the config-2 frame (8-row strips
dealt round-robin) timed in turn, world = 4 and 8, 20 and 32 frames per launch.  max / mean is what the slowest rank
costs the job (the gather is not included)."""
import os, statistics, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
W, H, N = 1920, 1080, 64
arrays = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
tr = rt.RayTracer(0, W, H)
tr.load_scene(arrays)
for batch in (20, 32):
    tr.set_option("batch_frames", batch)
    for world in (4, 8):
        res = []
        for rank in range(world):
            ts = []
            for r in range(4):
                tr.reset_timing()
                tr.render_strips_frames(rt.make_params(W, H, 4, 8, skybox=1, frames=1), batch * 2, rank, world)
                st = tr.stats()
                if r:
                    ts.append(st.kernel_ms / st.frames)
            res.append(statistics.median(ts))
        print(f"batch {batch} world {world}: " + " ".join(f"{t:.4f}" for t in res) + f"  max {max(res):.4f} mean {sum(res)/len(res):.4f} ratio {max(res)/(sum(res)/len(res)):.3f}", flush=True)
