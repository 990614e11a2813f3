import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from oracle import oracle
# a floor quad (root leaf) + an emissive quad above it with its own transform + many-mesh kernels forced
sc = rt.Scene()
sc.set_camera((0, 1.0, 4.0), (0, 0.5, 0), fov=50.0)
quad = [[-1, 0, -1, 0, 1, 0, 0, 0], [1, 0, -1, 0, 1, 0, 1, 0], [1, 0, 1, 0, 1, 0, 1, 1], [-1, 0, 1, 0, 1, 0, 0, 1]]
sc.add_mesh_from_data(quad, [2, 1, 0, 3, 2, 0], xform=rt.transform(scale=(4, 1, 4)), mat=rt.material(color=(0.8, 0.8, 0.7, 1), smoothness=0.0))
h = float(np.sin(np.pi / 4))
sc.add_mesh_from_data(quad, [0, 1, 2, 0, 2, 3], xform=rt.transform(pos=(0, 3, 0), scale=(3, 1, 3)),
                      mat=rt.material(color=(0.7, 0.7, 0.7, 1), emission_color=(1, 1, 1, 1), emission_strength=4.0))
sc.build()
a = rt.SceneArrays.from_scene(sc)
w, hh = 64, 36
tr = rt.RayTracer(0, w, hh)
tr.set_option("cull_roots", 1)
tr.load_scene(a)
for nb, spp in ((0, 1), (1, 1), (3, 2)):
    p = rt.make_params(w, hh, nb, spp, skybox=1, frames=0)
    ref, st = oracle.render(p, a)
    for wf in (0, 1):
        tr.set_option("wavefront", wf)
        tr.set_counters(True)
        tr.reset_timing()
        tr.render(p)
        got, s = tr.read_image(w, hh), tr.stats()
        bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
        print(f"two quads nb {nb} spp {spp} wf {wf}: bad {len(bad)} seg {s.segments}/{st.segments} tt {s.triangle_tests}/{st.triangle_tests} {tr.last_launch()['wavefront']} {tr.last_launch()['many_mesh']}", flush=True)
        for y, x in bad[:4]:
            rgba, rec = oracle.trace_pixel(p, a, int(x), int(y))
            print("  ", x, y, [(int(r[0]), int(r[1]), float(r[2])) for r in rec][:3], got[y, x], ref[y, x])
