"""`RayTracer`: Python face of the device-side C ABI, with the reference's
method names (src/rendering/ray_tracer.rs:48-435).

    rt = RayTracer(device=0, max_width=1920, max_height=1080)   # new + create_gpu_resources
    rt.load_scene_gpu_resources(arrays)                        # textures
    rt.update_buffers(arrays)                                  # scene arrays (on change)
    rt.render(params)                                          # one frame (async)
    img = rt.read_image(w, h)                                  # RGBA32F, row 0 = bottom

The render path is the HIP library only; nothing here computes pixels.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from .lib import RtError, load
from .scene import Scene, SceneArrays


class RayTracer:
    def __init__(self, device=0, max_width=1920, max_height=1080, lib=None):
        self._L = lib or load()   # (lib=load_test(): a handle of the test library, for the rt_test_* entry points)
        self._h = C.c_void_p()
        rc = self._L.rt_create(device, max_width, max_height, C.byref(self._h))
        if rc < 0:
            msg = self._L.rt_last_error(self._h).decode()
            if self._h:
                self._L.rt_destroy(self._h)
                self._h = None
            raise RtError(rc, msg)
        self.max_width, self.max_height = max_width, max_height
        # tuning knobs for experiments (results never depend on them): RT2_OPTIONS="forest=0,vote_eighths=5"
        for kv in os.environ.get("RT2_OPTIONS", "").split(","):
            if kv.strip():
                k, v = kv.split("=")
                self.set_option(k.strip(), int(v))

    def _check(self, rc):
        if rc < 0:
            raise RtError(rc, self._L.rt_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._L.rt_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- reference-named methods --------------------------------------
    def load_scene_gpu_resources(self, arrays):
        descs, n = arrays.texture_descs()
        self._check(self._L.rt_upload_textures(self._h, descs, n))

    def update_buffers(self, arrays):
        a = arrays
        u = a.uniform
        self._check(self._L.rt_upload_scene(
            self._h, C.byref(u), a.spheres.ctypes.data, a.spheres.shape[0], a.meshes.ctypes.data,
            a.meshes.shape[0], a.triangles.ctypes.data, a.triangles.shape[0], a.nodes.ctypes.data,
            a.nodes.shape[0]))

    def load_built_scene(self, scene):
        """A built Scene (C++ object) straight to the device: textures + arrays, without the round trip through numpy
        (rt_upload_built_scene: what the C++ mirror's load_scene_gpu_resources + update_buffers do)."""
        self._check(self._L.rt_upload_built_scene(self._h, scene._p))

    def load_scene(self, scene):
        """Scene (C++ object) or SceneArrays -> device."""
        arrays = SceneArrays.from_scene(scene) if isinstance(scene, Scene) else scene
        self.load_scene_gpu_resources(arrays)
        self.update_buffers(arrays)
        return arrays

    def set_camera(self, camera_uniform):
        self._check(self._L.rt_set_camera(self._h, C.byref(camera_uniform)))

    def render(self, params):
        self._check(self._L.rt_render(self._h, C.byref(params)))

    def render_frames(self, params, n_frames):
        """n_frames consecutive frames (Params.frames advancing), sampled in batches of overlapped frames."""
        self._check(self._L.rt_render_frames(self._h, C.byref(params), n_frames))

    def render_strips(self, params, rank, world):
        self._check(self._L.rt_render_strips(self._h, C.byref(params), rank, world))

    def render_strips_frames(self, params, n_frames, rank, world):
        self._check(self._L.rt_render_strips_frames(self._h, C.byref(params), n_frames, rank, world))

    # ---- data movement / bookkeeping ----------------------------------
    def synchronize(self):
        self._check(self._L.rt_synchronize(self._h))

    def read_image(self, width, height):
        out = np.empty((height, width, 4), dtype=np.float32)
        self._check(self._L.rt_read_image(self._h, out.ctypes.data, out.nbytes))
        return out

    def read_texels(self, n_texels):
        out = np.empty((n_texels, 4), dtype=np.float32)
        self._check(self._L.rt_read_image(self._h, out.ctypes.data, out.nbytes))
        return out

    def snapshot_image(self, width, height):
        """Keep the image as it is now aside (stream-ordered device copy; does not block)."""
        self._check(self._L.rt_snapshot_image(self._h, width * height * 16))

    def read_snapshot(self, width, height):
        """The last snapshot, read on a stream of its own: later frames are not waited for."""
        out = np.empty((height, width, 4), dtype=np.float32)
        self._check(self._L.rt_read_snapshot(self._h, out.ctypes.data, out.nbytes))
        return out

    def write_image(self, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        self._check(self._L.rt_write_image(self._h, img.ctypes.data, img.nbytes))

    def assemble_strips(self, gathered_device_ptr, width, height, world):
        self._check(self._L.rt_assemble_strips(self._h, gathered_device_ptr, width, height, world))

    def set_counters(self, enabled):
        self._check(self._L.rt_set_counters(self._h, int(enabled)))

    def set_option(self, name, value):
        self._check(self._L.rt_set_option(self._h, name.encode(), int(value)))

    def reset_timing(self):
        self._check(self._L.rt_reset_timing(self._h))

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.rt_set_stream(self._h, hip_stream_ptr))

    def bind_image(self, device_ptr, texels):
        self._check(self._L.rt_bind_image(self._h, device_ptr, texels))

    # ---- test-only: the device's arithmetic building blocks (tests/test_gpu_device_units.py) ----
    def sweep(self, which):
        """(floats checked, mismatches, a mismatching bit pattern) of the kernels' short reciprocal (which = 0) / square root
        (1) against the compiler's IEEE 1.0f / x / sqrt on the device, over every float the short form serves."""
        out = (C.c_uint64 * 3)()
        self._check(self._L.rt_test_sweep(self._h, which, out))
        return int(out[0]), int(out[1]), int(out[2])

    def device_units(self, fn, x, y=None):
        x = np.ascontiguousarray(x).view(np.float32).ravel()
        y = np.zeros_like(x) if y is None else np.ascontiguousarray(y).view(np.float32).ravel()
        out = np.empty_like(x)
        self._check(self._L.rt_test_device_units(self._h, fn, x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size))
        return out

    def device_sample_texture(self, tex_rgba8, uv):
        tex = np.ascontiguousarray(tex_rgba8, dtype=np.uint8)
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        d = A.TextureDesc(tex.ctypes.data, tex.shape[1], tex.shape[0])
        out = np.empty((uv.shape[0], 4), np.float32)
        self._check(self._L.rt_test_device_sample_texture(self._h, C.byref(d), uv.ctypes.data, out.ctypes.data, uv.shape[0]))
        return out

    def stats(self):
        s = A.Stats()
        self._check(self._L.rt_get_stats(self._h, C.byref(s)))
        return s

    def last_launch(self):
        """Shape of the last render launch: dynamic LDS bytes per workgroup, workgroups, scene staged in LDS, kernel flags."""
        out = (C.c_uint32 * 6)()
        self._check(self._L.rt_last_launch(self._h, C.byref(out)))
        return {"lds_bytes_per_workgroup": out[0], "workgroups": out[1], "scene_in_lds": bool(out[2]),
                "many_mesh": bool(out[3] & 1), "specialised": bool(out[3] & 2), "one_wave_per_tile": bool(out[3] & 4),
                "deferred_walks": bool(out[3] & 8), "wavefront": bool(out[3] & 16),
                "device_mb_held": out[4], "device_mb_cap": out[5]}

    @property
    def device_image_ptr(self):
        return self._L.rt_device_image(self._h)

    @property
    def stream_ptr(self):
        return self._L.rt_stream(self._h)

    def strip_texels(self, width, height, rank, world):
        return int(self._L.rt_strip_texels(width, height, rank, world))


def render_multi(tracers, params, read_back=True, n_frames=1):
    """rt_render_multi(_frames) over a list of RayTracer (one per device, same scene on each):
    returns the assembled frame (H, W, 4) f32 when read_back, else None (non-blocking)."""
    L = load()
    arr = (C.c_void_p * len(tracers))(*[t._h for t in tracers])
    out = np.empty((params.height, params.width, 4), np.float32) if read_back else None
    rc = L.rt_render_multi_frames(arr, len(tracers), C.byref(params), n_frames, out.ctypes.data if read_back else None)
    if rc < 0:
        raise RtError(rc, L.rt_last_error(tracers[0]._h).decode())
    return out


def read_multi_frame(root, width, height):
    """The frame the last render_multi call assembled on the root (blocking)."""
    L = load()
    out = np.empty((height, width, 4), np.float32)
    rc = L.rt_read_multi_frame(root._h, out.ctypes.data, out.nbytes)
    if rc < 0:
        raise RtError(rc, L.rt_last_error(root._h).decode())
    return out
