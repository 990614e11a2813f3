"""Loader and prototypes of the C-ABI product library (include/rt_abi.h).

The library is mandatory: there is no Python or CPU fallback for the render
path.  Importing this module on a machine without the built .so raises.
"""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT2_LIB lets tuning tools load an experimental build; the product is librt2_mi355x.so
LIB_PATH = os.environ.get("RT2_LIB") or os.path.join(_HERE, "librt2_mi355x.so")
# The test library: the same sources and flags plus -DRT_TEST_ENTRIES=1 -- everything the product exports and the test-only
# entry points of include/rt_test_abi.h (round 5: the product library no longer carries them).
TEST_LIB_PATH = os.path.join(_HERE, "librt2_mi355x_test.so")

# every symbol include/rt_abi.h declares
EXPORTS = [
    "rt_create", "rt_upload_scene", "rt_upload_textures", "rt_set_camera", "rt_render",
    "rt_render_strips", "rt_render_multi", "rt_render_frames", "rt_render_strips_frames", "rt_render_multi_frames", "rt_read_multi_frame", "rt_strip_texels", "rt_assemble_strips", "rt_read_image", "rt_write_image", "rt_snapshot_image", "rt_read_snapshot",
    "rt_synchronize", "rt_get_stats", "rt_last_launch", "rt_reset_timing", "rt_bind_image", "rt_set_stream", "rt_set_option", "rt_set_counters", "rt_device_image", "rt_stream",
    "rt_last_error", "rt_destroy", "rt_version", "rt_device_count", "rt_abi_sizes",
    "rt_scene_load_builtin", "rt_scene_create", "rt_scene_set_camera", "rt_transform_cam",
    "rt_scene_add_sphere", "rt_scene_add_obj", "rt_scene_add_mesh_data",
    "rt_scene_add_texture_rgba8", "rt_scene_build", "rt_scene_build_device", "rt_scene_get_uniform",
    "rt_scene_num_spheres", "rt_scene_num_meshes", "rt_scene_num_triangles", "rt_scene_num_nodes",
    "rt_scene_num_textures", "rt_scene_spheres", "rt_scene_meshes", "rt_scene_triangles",
    "rt_scene_nodes", "rt_scene_get_texture", "rt_scene_mesh_label", "rt_scene_mesh_data", "rt_scene_num_mesh_instances", "rt_scene_last_error",
    "rt_scene_destroy", "rt_upload_built_scene", "rt_scene_subdivide_meshes", "rt_export_rgba8",
]
# every symbol include/rt_test_abi.h declares (the test library only)
TEST_EXPORTS = ["rt_test_device_units", "rt_test_sweep", "rt_test_device_sample_texture", "rt_test_read_wavefront",
                "rt_test_rccl_gather", "rt_test_frame_ahead_depth"]

_lib = None
_test_lib = None


def load_test():
    """The test library (include/rt_test_abi.h beside include/rt_abi.h).  A handle belongs to the library that made it:
    RayTracer(lib=load_test()) drives a handle and the test entry points through this one."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            from .build import build_test_library
            build_test_library()
        _test_lib = _bind(C.CDLL(TEST_LIB_PATH), with_test_entries=True)
    return _test_lib


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("RT2_LIB"):
        # Not a fallback: build the one and only implementation (hipcc cross-compiles gfx950).
        try:
            from .build import build_product
            build_product()
        except Exception as e:  # noqa: BLE001
            raise ImportError(f"{LIB_PATH} is missing and could not be built: {e}") from e
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). There is no fallback path.")
    _lib = _bind(C.CDLL(LIB_PATH), with_test_entries=False)
    return _lib


def _bind(L, with_test_entries):
    P = C.POINTER
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "rt_create": (i32, [i32, u32, u32, P(vp)]),
        "rt_upload_scene": (i32, [vp, P(A.SceneUniform), vp, u32, vp, u32, vp, u32, vp, u32]),
        "rt_upload_textures": (i32, [vp, P(A.TextureDesc), u32]),
        "rt_set_camera": (i32, [vp, P(A.CameraUniform)]),
        "rt_render": (i32, [vp, P(A.Params)]),
        "rt_render_strips": (i32, [vp, P(A.Params), u32, u32]),
        "rt_render_multi": (i32, [P(vp), i32, P(A.Params), vp]),
        "rt_render_frames": (i32, [vp, P(A.Params), u32]),
        "rt_render_strips_frames": (i32, [vp, P(A.Params), u32, u32, u32]),
        "rt_render_multi_frames": (i32, [P(vp), i32, P(A.Params), u32, vp]),
        "rt_read_multi_frame": (i32, [vp, vp, C.c_size_t]),
        "rt_strip_texels": (u64, [u32, u32, u32, u32]),
        "rt_assemble_strips": (i32, [vp, vp, u32, u32, u32]),
        "rt_read_image": (i32, [vp, vp, C.c_size_t]),
        "rt_write_image": (i32, [vp, vp, C.c_size_t]),
        "rt_snapshot_image": (i32, [vp, C.c_size_t]),
        "rt_read_snapshot": (i32, [vp, vp, C.c_size_t]),
        "rt_synchronize": (i32, [vp]),
        "rt_get_stats": (i32, [vp, P(A.Stats)]),
        "rt_last_launch": (i32, [vp, P(u32 * 6)]),
        "rt_set_counters": (i32, [vp, i32]),
        "rt_set_option": (i32, [vp, C.c_char_p, i32]),
        "rt_reset_timing": (i32, [vp]),
        "rt_bind_image": (i32, [vp, vp, u64]),
        "rt_set_stream": (i32, [vp, vp]),
        "rt_device_image": (vp, [vp]),
        "rt_stream": (vp, [vp]),
        "rt_last_error": (C.c_char_p, [vp]),
        "rt_destroy": (None, [vp]),
        "rt_version": (C.c_char_p, []),
        "rt_device_count": (i32, []),
        "rt_abi_sizes": (None, [P(u32 * 8)]),
        "rt_scene_load_builtin": (i32, [C.c_char_p, C.c_char_p, P(vp)]),
        "rt_scene_create": (i32, [P(vp)]),
        "rt_scene_set_camera": (i32, [vp, P(A.CameraDesc)]),
        "rt_transform_cam": (None, [P(C.c_float * 3), P(C.c_float * 3), P(A.Transform)]),
        "rt_scene_add_sphere": (i32, [vp, P(C.c_float * 3), C.c_float, P(A.Material)]),
        "rt_scene_add_obj": (i32, [vp, C.c_char_p, C.c_char_p, P(A.Transform), i32, P(A.Material)]),
        "rt_scene_add_mesh_data": (i32, [vp, vp, u32, vp, u32, P(A.Transform), P(A.Material)]),
        "rt_scene_add_texture_rgba8": (i32, [vp, vp, u32, u32]),
        "rt_scene_build": (i32, [vp, i32]),
        "rt_scene_build_device": (i32, [vp, i32, i32, u32]),
        "rt_scene_get_uniform": (i32, [vp, P(A.SceneUniform)]),
        "rt_scene_num_spheres": (u32, [vp]),
        "rt_scene_num_meshes": (u32, [vp]),
        "rt_scene_num_triangles": (u32, [vp]),
        "rt_scene_num_nodes": (u32, [vp]),
        "rt_scene_num_textures": (u32, [vp]),
        "rt_scene_spheres": (vp, [vp]),
        "rt_scene_meshes": (vp, [vp]),
        "rt_scene_triangles": (vp, [vp]),
        "rt_scene_nodes": (vp, [vp]),
        "rt_scene_get_texture": (i32, [vp, u32, P(A.TextureDesc)]),
        "rt_scene_mesh_label": (C.c_char_p, [vp, u32]),
        "rt_scene_mesh_data": (i32, [vp, u32, vp, P(u32), vp, P(u32), P(A.Transform), P(A.Material)]),
        "rt_scene_num_mesh_instances": (u32, [vp]),
        "rt_scene_last_error": (C.c_char_p, [vp]),
        "rt_scene_destroy": (None, [vp]),
        "rt_upload_built_scene": (i32, [vp, vp]),
        "rt_scene_subdivide_meshes": (i32, [vp, u32]),
        "rt_export_rgba8": (i32, [vp, u32, u32, vp]),
    }
    assert set(sig) == set(EXPORTS)
    if with_test_entries:
        sig.update({
            "rt_test_device_units": (i32, [vp, i32, vp, vp, vp, u64]),
            "rt_test_sweep": (i32, [vp, i32, vp]),
            "rt_test_device_sample_texture": (i32, [vp, P(A.TextureDesc), vp, vp, u64]),
            "rt_test_frame_ahead_depth": (i32, [i32, u64, i32, i32, i32]),
            "rt_test_read_wavefront": (i32, [vp, i32, vp, u64]),
            "rt_test_rccl_gather": (i32, [C.c_char_p, i32]),
        })
        assert set(sig) == set(EXPORTS) | set(TEST_EXPORTS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # raises AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    sizes = (u32 * 8)()
    L.rt_abi_sizes(C.byref(sizes))
    expect = [C.sizeof(t) for t in (A.Params, A.Material, A.Sphere, A.MeshUniform, A.Node,
                                    A.PackedTriangle, A.CameraUniform, A.SceneUniform)]
    if list(sizes) != expect:
        raise ImportError(f"ABI mismatch: library {list(sizes)} vs bindings {expect}")
    return L


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code
