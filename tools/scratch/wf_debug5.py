import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from oracle import oracle
sc = rt.Scene()
sc.set_camera((0, 1.0, 4.0), (0, 0.5, 0), fov=50.0)
quad = [[-1, 0, -1, 0, 1, 0, 0, 0], [1, 0, -1, 0, 1, 0, 1, 0], [1, 0, 1, 0, 1, 0, 1, 1], [-1, 0, 1, 0, 1, 0, 0, 1]]
sc.add_mesh_from_data(quad, [2, 1, 0, 3, 2, 0], xform=rt.transform(scale=(4, 1, 4)), mat=rt.material(color=(0.8, 0.8, 0.7, 1), smoothness=0.0))
sc.add_mesh_from_data(quad, [0, 1, 2, 0, 2, 3], xform=rt.transform(pos=(0, 3, 0), scale=(3, 1, 3)),
                      mat=rt.material(color=(0.7, 0.7, 0.7, 1), emission_color=(1, 1, 1, 1), emission_strength=4.0))
sc.build()
a = rt.SceneArrays.from_scene(sc)
w, hh = 64, 36
tr = rt.RayTracer(0, w, hh)
tr.set_option("cull_roots", 1)
tr.set_option("wavefront", 1)
tr.load_scene(a)
p = rt.make_params(w, hh, 0, 1, skybox=1, frames=0)
ref, st = oracle.render(p, a)
tr.render(p)
got = tr.read_image(w, hh)
L = rt.load()
nslots = 8 * 5 * 64
blocks = nslots // 64
hit = np.zeros((blocks, 2, 64, 4), np.float32)
L.rt_test_read_wavefront(tr._h, 1, hit.ctypes.data, hit.nbytes)
state = np.zeros((blocks, 6, 64, 4), np.float32)
L.rt_test_read_wavefront(tr._h, 0, state.ctypes.data, state.nbytes)
counts = np.zeros(3, np.uint32)
L.rt_test_read_wavefront(tr._h, 3, counts.ctypes.data, counts.nbytes)
print("counts", counts)
bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
print("bad", len(bad))
# slot of pixel (x, y): tile-major
def slot_of(x, y):
    tiles_x = (w + 7) // 8
    return ((y >> 3) * tiles_x + (x >> 3)) * 64 + ((y & 7) << 3 | (x & 7))
for (y, x) in list(bad[:3]) + [(5, 30), (20, 30)]:
    s = slot_of(int(x), int(y))
    h0, h1 = hit[s >> 6, 0, s & 63], hit[s >> 6, 1, s & 63]
    st0 = state[s >> 6, :, s & 63]
    rgba, rec = oracle.trace_pixel(p, a, int(x), int(y))
    print("pixel", x, y, "slot", s, "oracle", [(int(r[0]), int(r[1]), float(r[2])) for r in rec], "hit rec", h0[0], hex(h0.view(np.uint32)[1]), h0[2:], h1.view(np.uint32)[0], h1[1:], "xy", hex(st0[0].view(np.uint32)[0]), "ro", st0[1][:3], "rd", st0[1][3], st0[2][:2])
