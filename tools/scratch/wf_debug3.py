import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from oracle import oracle
a = rt.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"))
w, h = 96, 54
tr = rt.RayTracer(0, w, h)
for k, v in (("cull_roots", 1), ("forest", 0), ("flat2", 0)):
    tr.set_option(k, v)
tr.load_scene(a)
for nb, spp in ((0, 1), (1, 1), (4, 2)):
    p = rt.make_params(w, h, nb, spp, skybox=1, frames=0)
    ref, st = oracle.render(p, a)
    for wf in (0, 1):
        tr.set_option("wavefront", wf)
        tr.set_counters(True)
        tr.reset_timing()
        tr.render(p)
        got, s = tr.read_image(w, h), tr.stats()
        bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
        print(f"cornell nb {nb} spp {spp} wf {wf}: bad {len(bad)} seg {s.segments}/{st.segments} nt {s.node_tests}/{st.node_tests} tt {s.triangle_tests}/{st.triangle_tests} {tr.last_launch()}", flush=True)
        for y, x in bad[:6]:
            rgba, rec = oracle.trace_pixel(p, a, int(x), int(y))
            print("  ", x, y, [(int(r[0]), int(r[1]), float(r[2])) for r in rec][:3], got[y, x], ref[y, x])
