"""ctypes / numpy mirror of include/rt_abi.h (section 1: the data contract).

Every structure is byte-identical to the `#[repr(C)]` struct the reference
uploads to its shader (citations in rt_abi.h).  `check_sizes()` compares these
against the sizes the loaded C library reports (rt_abi_sizes).
"""
import ctypes as C

import numpy as np

f32, u32, i32 = C.c_float, C.c_uint32, C.c_int32


class Params(C.Structure):  # src/core/app.rs:27-40
    _fields_ = [("width", u32), ("height", u32), ("number_of_bounces", i32),
                ("rays_per_pixel", i32), ("skybox", i32), ("frames", i32),
                ("accumulate", i32), ("debug_flag", i32), ("debug_scale", i32),
                ("_p1", f32 * 3)]


class Material(C.Structure):  # src/scene/components/material.rs:3-18
    _fields_ = [("color", f32 * 4), ("emission_color", f32 * 4), ("specular_color", f32 * 4),
                ("absorption", f32 * 4), ("absorption_strength", f32),
                ("emission_strength", f32), ("smoothness", f32), ("specular", f32),
                ("ior", f32), ("flag", i32), ("diffuse_index", i32), ("normal_index", i32)]


class Sphere(C.Structure):  # geometry/sphere.rs:4-10
    _fields_ = [("pos", f32 * 3), ("radius", f32), ("material", Material)]


class MeshUniform(C.Structure):  # geometry/mesh.rs:52-62
    _fields_ = [("world_to_model", (f32 * 4) * 4), ("model_to_world", (f32 * 4) * 4),
                ("node_offset", u32), ("triangles", u32), ("triangle_offset", u32),
                ("_p1", f32), ("material", Material)]


class Node(C.Structure):  # src/core/bvh.rs:55-66
    _fields_ = [("left", u32), ("right", u32), ("first", u32), ("count", u32),
                ("aabb_min", f32 * 3), ("_p1", f32), ("aabb_max", f32 * 3), ("_p2", f32)]


class PackedTriangle(C.Structure):  # src/core/bvh.rs:19-34
    _fields_ = [("v1", f32 * 3), ("uv10", f32), ("v2", f32 * 3), ("uv11", f32),
                ("v3", f32 * 3), ("uv20", f32), ("n1", f32 * 3), ("uv21", f32),
                ("n2", f32 * 3), ("uv30", f32), ("n3", f32 * 3), ("uv31", f32)]


class CameraUniform(C.Structure):  # src/scene/camera.rs:15-22
    _fields_ = [("cam_to_world", (f32 * 4) * 4), ("view_params", f32 * 3),
                ("defocus_strength", f32), ("diverge_strength", f32)]


class SceneUniform(C.Structure):  # src/scene/scene.rs:1016-1026
    _fields_ = [("spheres", u32), ("n_vertices", u32), ("n_indices", u32), ("meshes", u32),
                ("camera", CameraUniform), ("nodes", u32), ("padding", f32 * 6)]


class TextureDesc(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", u32), ("height", u32)]


class Transform(C.Structure):  # components/transform.rs:3-8
    _fields_ = [("pos", f32 * 3), ("rot", f32 * 4), ("scale", f32 * 3)]

    @staticmethod
    def identity():
        return Transform((f32 * 3)(0, 0, 0), (f32 * 4)(0, 0, 0, 1), (f32 * 3)(1, 1, 1))


class CameraDesc(C.Structure):  # CameraDescriptor, camera.rs:38-66
    _fields_ = [("transform", Transform), ("fov", f32), ("aspect", f32), ("near_plane", f32),
                ("far_plane", f32), ("focus_dist", f32), ("defocus_strength", f32),
                ("diverge_strength", f32)]


class Stats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("paths", C.c_uint64), ("node_tests", C.c_uint64),
                ("triangle_tests", C.c_uint64), ("kernel_ms", f32), ("launches", u32), ("frames", u32), ("segments_reused", C.c_uint64),
                ("frames_speculative", u32), ("_reserved", u32)]


EXPECTED_SIZES = {Params: 48, Material: 96, Sphere: 112, MeshUniform: 240, Node: 48,
                  PackedTriangle: 96, CameraUniform: 84, SceneUniform: 128}
for _t, _s in EXPECTED_SIZES.items():
    assert C.sizeof(_t) == _s, (_t, C.sizeof(_t), _s)

# numpy views of the array types (for fixtures and inspection)
NODE_DTYPE = np.dtype([("left", "<u4"), ("right", "<u4"), ("first", "<u4"), ("count", "<u4"),
                       ("aabb_min", "<f4", 3), ("_p1", "<f4"), ("aabb_max", "<f4", 3),
                       ("_p2", "<f4")])
TRI_DTYPE = np.dtype([("v1", "<f4", 3), ("uv10", "<f4"), ("v2", "<f4", 3), ("uv11", "<f4"),
                      ("v3", "<f4", 3), ("uv20", "<f4"), ("n1", "<f4", 3), ("uv21", "<f4"),
                      ("n2", "<f4", 3), ("uv30", "<f4"), ("n3", "<f4", 3), ("uv31", "<f4")])
MATERIAL_DTYPE = np.dtype([("color", "<f4", 4), ("emission_color", "<f4", 4),
                           ("specular_color", "<f4", 4), ("absorption", "<f4", 4),
                           ("absorption_strength", "<f4"), ("emission_strength", "<f4"),
                           ("smoothness", "<f4"), ("specular", "<f4"), ("ior", "<f4"),
                           ("flag", "<i4"), ("diffuse_index", "<i4"), ("normal_index", "<i4")])
MESH_DTYPE = np.dtype([("world_to_model", "<f4", (4, 4)), ("model_to_world", "<f4", (4, 4)),
                       ("node_offset", "<u4"), ("triangles", "<u4"), ("triangle_offset", "<u4"),
                       ("_p1", "<f4"), ("material", MATERIAL_DTYPE)])
SPHERE_DTYPE = np.dtype([("pos", "<f4", 3), ("radius", "<f4"), ("material", MATERIAL_DTYPE)])
assert NODE_DTYPE.itemsize == 48 and TRI_DTYPE.itemsize == 96
assert MESH_DTYPE.itemsize == 240 and SPHERE_DTYPE.itemsize == 112

RT_OK = 0
MATERIAL_DEFAULT, MATERIAL_GLASS, MATERIAL_TEXTURE = 0, 1, 2


def make_params(width, height, bounces, spp, skybox=1, frames=0, accumulate=1, debug_flag=0,
                debug_scale=0):
    return Params(width, height, bounces, spp, skybox, frames, accumulate, debug_flag,
                  debug_scale, (f32 * 3)(0, 0, 0))
