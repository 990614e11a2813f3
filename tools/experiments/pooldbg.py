import os, sys
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import numpy as np
import ray_tracer_2_amd as rt
a = rt.SceneArrays.load(os.path.join(ROOT,"tests","golden","cornell_scene.npz"))
W=H=256
def fresh(**o):
    tr = rt.RayTracer(0, 1920, 1080); tr.load_scene(a); tr.set_option("kernel_variant", 0)
    for k,v in o.items(): tr.set_option(k,v)
    tr.render(rt.make_params(W,H,1,1,skybox=1,frames=0)); g = tr.read_image(W,H).copy()
    tr.render(rt.make_params(W,H,1,1,skybox=1,frames=0)); g2 = tr.read_image(W,H).copy()
    return g, g2
ref, ref2 = fresh(pool=0)
print("pool=0 first==second", np.array_equal(ref.view(np.uint32), ref2.view(np.uint32)))
for kw in (dict(pool=1), dict(pool=1, tile_feedback=0), dict(pool=1, pool_dense=9), dict(pool=1, pool_dense=0), dict(pool=1, pixel_cache=0)):
    g, g2 = fresh(**kw)
    for name, im in (("first", g), ("second", g2)):
        bad = np.argwhere((im.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
        msg = f"{kw} {name}: bad pixels {len(bad)}"
        if len(bad):
            ys, xs = bad[:,0], bad[:,1]
            msg += f" y {ys.min()}..{ys.max()} x {xs.min()}..{xs.max()}; e.g. {tuple(bad[0])} {im[tuple(bad[0])]} vs {ref[tuple(bad[0])]}"
        print(msg)
