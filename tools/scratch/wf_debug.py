import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
from oracle import oracle
a = rt.SceneArrays.from_scene(scenes.sponza_standin(200))
w, h = 192, 108
tr = rt.RayTracer(0, w, h)
tr.load_scene(a)
for (nb, spp) in ((0, 1), (1, 1), (4, 1), (4, 4)):
    p = rt.make_params(w, h, nb, spp, skybox=1, frames=0)
    ref, st = oracle.render(p, a)
    for c in (True, False):
        for wf in (0, 1):
            tr.set_option("wavefront", wf)
            tr.set_counters(c)
            tr.reset_timing()
            tr.render(p)
            got, s = tr.read_image(w, h), tr.stats()
            bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
            print(f"nb {nb} spp {spp} counters {c} wf {wf}: bad pixels {len(bad)} first {bad[:5].tolist()}  seg {s.segments} vs {st.segments}  nt {s.node_tests} vs {st.node_tests} tt {s.triangle_tests} vs {st.triangle_tests} launch {tr.last_launch()['wavefront']}", flush=True)
            if len(bad) and wf:
                y, x = bad[0]
                print("   got", got[y, x], "ref", ref[y, x])
