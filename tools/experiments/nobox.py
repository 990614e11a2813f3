import os, sys, statistics
ROOT="/root/repo" if os.path.exists("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
import numpy as np
import ray_tracer_2_amd as rt
W,H=1920,1080
a = rt.SceneArrays.load(os.path.join(ROOT,"tests","golden","cornell_scene.npz"))
def bench(arr, label):
    tr = rt.RayTracer(0, W, H); tr.load_scene(arr)
    ts=[]
    for r in range(5):
        tr.reset_timing()
        for f in range(4): tr.render(rt.make_params(W,H,4,8,skybox=1,frames=1+f))
        st=tr.stats()
        if r: ts.append(st.kernel_ms/st.launches)
    print(label, "%.3f ms" % statistics.median(ts), "rays/frame %.1fM" % (st.segments/st.launches/1e6))
bench(a, "full")
import copy
b = copy.copy(a)
b.meshes = a.meshes[[0,1,2,3,4,7]].copy()
b.uniform = copy.copy(a.uniform); b.uniform.meshes = 6
bench(b, "no boxes")
c = copy.copy(a)
c.meshes = a.meshes[[0,1,3,7,5,6]].copy()
c.uniform = copy.copy(a.uniform); c.uniform.meshes = 6
bench(c, "no 3-node quads")
