import os, sys, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt
from ray_tracer_2_amd import scenes
g=os.path.join(ROOT,"tests","golden")
t0=time.perf_counter()
sc = scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g,"cornell_raw.npz")), scenes.load_raw_meshes(os.path.join(g,"dragon_raw.npz")), subdivide=11, device=0)
t1=time.perf_counter()
a = rt.SceneArrays.from_scene(sc)
t2=time.perf_counter()
tr = rt.RayTracer(0, 1920, 1080)
t3=time.perf_counter()
tr.load_scene(a); tr.synchronize()
t4=time.perf_counter()
tr.load_scene(a); tr.synchronize()
t5=time.perf_counter()
print(f"scene+subdivide+build(gpu) {t1-t0:.2f} s; arrays {t2-t1:.2f} s; create {t3-t2:.2f} s; upload {t4-t3:.2f} s; upload again {t5-t4:.2f} s")
