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
p = rt.make_params(w, h, 1, 1, skybox=1, frames=0)
ref, st = oracle.render(p, a)
tr.set_option("wavefront", 1)
tr.render(p)
got = tr.read_image(w, h)
bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
print(len(bad))
import collections
c = collections.Counter()
for y, x in bad[:40]:
    rgba, rec = oracle.trace_pixel(p, a, int(x), int(y))
    print(x, y, [tuple(r) for r in rec][:3], got[y, x], ref[y, x])
