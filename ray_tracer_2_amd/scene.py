"""Host-side scene model: Python face of section 3 of include/rt_abi.h.

Mirrors the reference's `Scene` / `SceneDefinition` (src/scene/scene.rs:70-278):
`Scene.from_name("cornell_box", assets_dir)` ≙ Scene::from_name +
instantiate_scene; `add_sphere` / `add_mesh_from_file` / `add_mesh_from_data` /
`set_camera` ≙ SceneDefinition's builders; `build()` ≙ BVH::build_per_mesh.
All work happens in the C++ library.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from .lib import RtError, load

DEFAULT_ASSETS = os.environ.get("RT2_ASSETS_DIR", "/root/reference/assets")


def material(color=(0.7, 0.7, 0.7, 1.0), emission_color=(0, 0, 0, 0), specular_color=(0, 0, 0, 0),
             absorption=(0, 0, 0, 0), absorption_strength=0.0, emission_strength=0.0, smoothness=0.9,
             specular=0.0, ior=1.0, flag=A.MATERIAL_DEFAULT, diffuse_index=-1, normal_index=-1):
    """MaterialUniform with the defaults of MaterialUniform::default (material.rs:19-36)."""
    f4 = C.c_float * 4
    return A.Material(f4(*color), f4(*emission_color), f4(*specular_color), f4(*absorption),
                      absorption_strength, emission_strength, smoothness, specular, ior, flag,
                      diffuse_index, normal_index)


def transform(pos=(0, 0, 0), rot=(0, 0, 0, 1), scale=(1, 1, 1)):
    return A.Transform((C.c_float * 3)(*pos), (C.c_float * 4)(*rot), (C.c_float * 3)(*scale))


class Scene:
    def __init__(self, ptr=None):
        self._L = load()
        if ptr is None:
            ptr = C.c_void_p()
            self._check(self._L.rt_scene_create(C.byref(ptr)), None)
        self._p = ptr

    def _check(self, rc, p=None):
        if rc < 0:
            p = p if p is not None else getattr(self, "_p", None)
            msg = self._L.rt_scene_last_error(p).decode() if p else ""
            raise RtError(rc, msg)
        return rc

    @classmethod
    def from_name(cls, name, assets_dir=DEFAULT_ASSETS):
        L = load()
        ptr = C.c_void_p()
        rc = L.rt_scene_load_builtin(name.encode(), assets_dir.encode(), C.byref(ptr))
        if rc < 0:
            msg = L.rt_scene_last_error(ptr).decode() if ptr else ""
            if ptr:
                L.rt_scene_destroy(ptr)
            raise RtError(rc, msg)
        return cls(ptr)

    def close(self):
        if getattr(self, "_p", None):
            self._L.rt_scene_destroy(self._p)
            self._p = None

    __del__ = close

    # ---- SceneDefinition builders -------------------------------------
    def set_camera(self, origin, look_at, fov=90.0, aspect=16.0 / 9.0, near=0.01, far=1000.0,
                   focus_dist=1.0, defocus_strength=0.0, diverge_strength=0.0):
        t = A.Transform()
        self._L.rt_transform_cam(C.byref((C.c_float * 3)(*origin)), C.byref((C.c_float * 3)(*look_at)),
                                 C.byref(t))
        d = A.CameraDesc(t, fov, aspect, near, far, focus_dist, defocus_strength, diverge_strength)
        self._check(self._L.rt_scene_set_camera(self._p, C.byref(d)))

    def add_sphere(self, centre, radius, mat):
        self._check(self._L.rt_scene_add_sphere(self._p, C.byref((C.c_float * 3)(*centre)), radius,
                                                C.byref(mat)))

    def add_mesh_from_file(self, path, xform=None, use_mtl=True, mat=None, assets_dir=DEFAULT_ASSETS):
        xform = xform or transform()
        mat = mat or material()
        self._check(self._L.rt_scene_add_obj(self._p, assets_dir.encode(), path.encode(),
                                             C.byref(xform), int(use_mtl), C.byref(mat)))

    def add_mesh_from_data(self, vertices8, indices, xform=None, mat=None):
        v = np.ascontiguousarray(vertices8, dtype=np.float32).reshape(-1, 8)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        xform = xform or transform()
        mat = mat or material()
        self._check(self._L.rt_scene_add_mesh_data(self._p, v.ctypes.data, v.shape[0], i.ctypes.data,
                                                   i.shape[0], C.byref(xform), C.byref(mat)))

    def add_texture_rgba8(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w, _ = rgba.shape
        return self._check(self._L.rt_scene_add_texture_rgba8(self._p, rgba.ctypes.data, w, h))

    def subdivide_meshes(self, n):
        self._check(self._L.rt_scene_subdivide_meshes(self._p, n))

    def build(self, quality=1, device=None, min_triangles=0):
        if device is None:
            self._check(self._L.rt_scene_build(self._p, quality))
        else:  # SAH searches of large meshes on the GPU; same result
            self._check(self._L.rt_scene_build_device(self._p, quality, int(device), int(min_triangles)))

    # ---- the arrays the hot path consumes -----------------------------
    def uniform(self):
        u = A.SceneUniform()
        self._check(self._L.rt_scene_get_uniform(self._p, C.byref(u)))
        return u

    def _array(self, fn, count, dtype):
        n = count(self._p)
        if n == 0:
            return np.zeros(0, dtype=dtype)
        ptr = fn(self._p)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).copy()

    def spheres(self):
        return self._array(self._L.rt_scene_spheres, self._L.rt_scene_num_spheres, A.SPHERE_DTYPE)

    def meshes(self):
        return self._array(self._L.rt_scene_meshes, self._L.rt_scene_num_meshes, A.MESH_DTYPE)

    def triangles(self):
        return self._array(self._L.rt_scene_triangles, self._L.rt_scene_num_triangles, A.TRI_DTYPE)

    def nodes(self):
        return self._array(self._L.rt_scene_nodes, self._L.rt_scene_num_nodes, A.NODE_DTYPE)

    def textures(self):
        out = []
        for i in range(self._L.rt_scene_num_textures(self._p)):
            d = A.TextureDesc()
            self._check(self._L.rt_scene_get_texture(self._p, i, C.byref(d)))
            n = d.width * d.height * 4
            buf = (C.c_uint8 * n).from_address(d.rgba8) if n else b""
            out.append(np.frombuffer(buf, dtype=np.uint8).reshape(d.height, d.width, 4).copy())
        return out

    def raw_meshes(self):
        """[(label, vertices8 [n, 8] f32, indices u32, Transform, Material)] in instance order."""
        out = []
        for i in range(self._L.rt_scene_num_mesh_instances(self._p)):
            nv, ni = C.c_uint32(), C.c_uint32()
            self._check(self._L.rt_scene_mesh_data(self._p, i, None, C.byref(nv), None, C.byref(ni), None, None))
            v = np.empty((nv.value, 8), np.float32)
            idx = np.empty(ni.value, np.uint32)
            t, m = A.Transform(), A.Material()
            self._check(self._L.rt_scene_mesh_data(self._p, i, v.ctypes.data, None, idx.ctypes.data, None,
                                                   C.byref(t), C.byref(m)))
            out.append((self._L.rt_scene_mesh_label(self._p, i).decode(), v, idx, t, m))
        return out

    def mesh_labels(self):
        return [self._L.rt_scene_mesh_label(self._p, i).decode()
                for i in range(self._L.rt_scene_num_meshes(self._p))]


class SceneArrays:
    """Plain-array form of a scene (what crosses the FFI): usable as a fixture
    (save/load .npz) and as input to both the HIP library and the oracle."""

    def __init__(self, uniform, spheres, meshes, triangles, nodes, textures=()):
        self.uniform = uniform
        self.spheres = np.ascontiguousarray(spheres)
        self.meshes = np.ascontiguousarray(meshes)
        self.triangles = np.ascontiguousarray(triangles)
        self.nodes = np.ascontiguousarray(nodes)
        self.textures = [np.ascontiguousarray(t) for t in textures]

    @classmethod
    def from_scene(cls, scene):
        return cls(scene.uniform(), scene.spheres(), scene.meshes(), scene.triangles(), scene.nodes(),
                   scene.textures())

    def save(self, path):
        u = np.frombuffer(bytes(self.uniform), dtype=np.uint8)
        tex = {f"texture_{i}": t for i, t in enumerate(self.textures)}
        np.savez_compressed(path, uniform=u, spheres=self.spheres, meshes=self.meshes,
                            triangles=self.triangles, nodes=self.nodes, **tex)

    @classmethod
    def load(cls, path):
        z = np.load(path)
        u = A.SceneUniform.from_buffer_copy(z["uniform"].tobytes())
        tex = []
        while f"texture_{len(tex)}" in z:
            tex.append(z[f"texture_{len(tex)}"])
        return cls(u, z["spheres"].astype(A.SPHERE_DTYPE), z["meshes"].astype(A.MESH_DTYPE),
                   z["triangles"].astype(A.TRI_DTYPE), z["nodes"].astype(A.NODE_DTYPE), tex)

    def texture_descs(self):
        n = len(self.textures)
        arr = (A.TextureDesc * max(n, 1))()
        for i, t in enumerate(self.textures):
            arr[i] = A.TextureDesc(t.ctypes.data, t.shape[1], t.shape[0])
        return arr, n
