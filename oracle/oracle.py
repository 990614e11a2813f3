"""ctypes face of oracle/_build/librt_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  PARITY UNPINNED (see shader_oracle.cpp's header).
"""
import ctypes as C
import os

import numpy as np

from ray_tracer_2_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "librt_oracle.so")


class OracleStats(C.Structure):
    _fields_ = [("segments", C.c_uint64), ("node_tests", C.c_uint64), ("triangle_tests", C.c_uint64)]


TRANSCRIPT_DTYPE = np.dtype([("hit_mesh", "<i4"), ("hit_tri", "<i4"), ("dst", "<f4"),
                             ("rng_after", "<u4")])
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            from ray_tracer_2_amd.build import build_oracle
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        L.oracle_render.restype = C.c_int
        L.oracle_render_rows.restype = C.c_int
        L.oracle_trace_pixel.restype = C.c_int
        L.oracle_next_random_number.restype = C.c_uint32
        L.oracle_next_random_number.argtypes = [C.POINTER(C.c_uint32)]
        L.oracle_rand.restype = C.c_float
        L.oracle_rand.argtypes = [C.POINTER(C.c_uint32)]
        L.oracle_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def hardware_threads():
    """Threads worth starting: the affinity mask capped by the cgroup CPU quota."""
    n = load().oracle_hardware_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _scene_args(arrays):
    descs, n = arrays.texture_descs()
    return (C.byref(arrays.uniform), C.c_void_p(arrays.spheres.ctypes.data),
            C.c_void_p(arrays.meshes.ctypes.data), C.c_void_p(arrays.triangles.ctypes.data),
            C.c_void_p(arrays.nodes.ctypes.data), descs, C.c_uint32(n))


def render(params, arrays, image=None, rows=None, threads=None):
    """One frame of wgsl `main` on the CPU.  `image` (H, W, 4) f32 is the
    accumulation target (read when params.frames >= 1).  Returns (image, stats)."""
    L = load()
    h, w = params.height, params.width
    if image is None:
        image = np.zeros((h, w, 4), dtype=np.float32)
    assert image.dtype == np.float32 and image.shape == (h, w, 4) and image.flags.c_contiguous
    st = OracleStats()
    threads = threads or hardware_threads()
    if rows is None:
        rows = (0, h)
    if isinstance(rows, tuple):
        row_list = np.arange(rows[0], min(rows[1], h), dtype=np.uint32)
    else:
        row_list = np.ascontiguousarray(rows, dtype=np.uint32)
    rc = L.oracle_render_rows(C.byref(params), *_scene_args(arrays), C.c_void_p(image.ctypes.data),
                              C.c_void_p(row_list.ctypes.data), C.c_uint32(row_list.size),
                              C.c_int(threads), C.byref(st))
    assert rc == 0
    return image, st


def trace_pixel(params, arrays, x, y, cap=64):
    L = load()
    rec = np.zeros(cap, dtype=TRANSCRIPT_DTYPE)
    rgba = (C.c_float * 4)()
    n = L.oracle_trace_pixel(C.byref(params), *_scene_args(arrays), C.c_uint32(x), C.c_uint32(y), rgba,
                             C.c_void_p(rec.ctypes.data), C.c_uint32(cap))
    return np.array(rgba, dtype=np.float32), rec[:n]


def max_stack_index(reset=True):
    """Largest BVH stack_index seen by renders run with threads=1 since the last reset."""
    L = load()
    L.oracle_max_stack_index.restype = C.c_uint32
    return int(L.oracle_max_stack_index(C.c_int(1 if reset else 0)))


def census(enable):
    """Census of triangle hits against the entry distance of their leaf boxes (shader_oracle.cpp census_hit; the
    hypothesis behind the product's cross-mesh pruning).  census(True) switches it on and clears it;
    census(False) switches it off and returns what was counted since."""
    L = load()
    out = (C.c_double * 7)()
    L.oracle_census(C.c_int(1 if enable else 0), out)
    keys = ("hits", "entry_gt_t", "gt_1e-6", "gt_1e-4", "gt_1pct", "gt_12.5pct", "max_ratio")
    return dict(zip(keys, out))


def rng_sequence(seed, n):
    L = load()
    s = C.c_uint32(seed)
    return np.array([L.oracle_next_random_number(C.byref(s)) for _ in range(n)], dtype=np.uint32)


def rand_sequence(seed, n):
    L = load()
    s = C.c_uint32(seed)
    return np.array([L.oracle_rand(C.byref(s)) for _ in range(n)], dtype=np.float32)


FN = {"log": 0, "cos": 1, "sin": 2, "exp": 3, "exp2": 4, "log2": 5, "pow": 6, "acos": 7,
      "atan2": 8, "sqrt": 9, "div": 10, "rand": 11, "rng": 12, "trig_signbits": 13, "rand_normal_dist": 14,
      "rand_convert": 15, "normalize_x": 16}


def transc(fn, x, y=None):
    L = load()
    # (integer inputs -- RNG states, raw generator outputs -- travel as bit patterns)
    x = np.ascontiguousarray(x)
    x = x.view(np.float32) if x.dtype == np.uint32 else x.astype(np.float32, copy=False)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float32)
    out = np.empty_like(x)
    L.oracle_transc(C.c_int(FN[fn]), C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data),
                    C.c_void_p(out.ctypes.data), C.c_size_t(x.size))
    return out


def export_rgba8(img):
    """save_render_to_file's pixel loop (app.rs:408-460) on an (H, W, 4) f32 image -> (H, W, 4) u8."""
    L = load()
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    out = np.empty((h, w, 4), np.uint8)
    L.oracle_export_rgba8(C.c_void_p(img.ctypes.data), C.c_uint32(w), C.c_uint32(h), C.c_void_p(out.ctypes.data))
    return out


def sample_texture(tex_rgba8, uv):
    L = load()
    tex = np.ascontiguousarray(tex_rgba8, dtype=np.uint8)
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    d = A.TextureDesc(tex.ctypes.data, tex.shape[1], tex.shape[0])
    out = np.empty((uv.shape[0], 4), dtype=np.float32)
    L.oracle_sample_texture(C.byref(d), C.c_void_p(uv.ctypes.data), C.c_void_p(out.ctypes.data),
                            C.c_size_t(uv.shape[0]))
    return out
