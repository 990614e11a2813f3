"""The C++ mirror of the reference's RayTracer / Params (csrc/host/ray_tracer.hpp), driven the way the
reference's App drives the Rust struct (tests/cpp/ray_tracer_class_driver.cpp), against the ctypes path
over the same frames.  The driver is a library called in-process: a process that has initialised the GPU
must not exec another program on this pool."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, bits

pytestmark = pytest.mark.gpu
DATA = os.path.join(ROOT, "tests", "data")


@pytest.mark.parametrize("name", ["room", "balls"])
def test_cpp_ray_tracer_class_renders_what_the_c_abi_renders(rt, oracle, name):
    from ray_tracer_2_amd.build import CLASS_DRIVER_SO, build_class_driver
    # built by __graft_entry__.build(); it travels with the snapshot like the product library
    lib = C.CDLL(CLASS_DRIVER_SO if os.path.exists(CLASS_DRIVER_SO) else build_class_driver())
    w, h, nb, spp, frames = 160, 90, 4, 2, 3
    out = np.zeros((h, w, 4), np.float32)
    seg = C.c_ulonglong(0)
    err = C.create_string_buffer(512)
    lib.rt2_class_driver.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.POINTER(C.c_ulonglong), C.c_char_p, C.c_size_t]
    rc = lib.rt2_class_driver(name.encode(), DATA.encode(), w, h, nb, spp, frames, out.ctypes.data, C.byref(seg), err, 512)
    assert rc == 0, err.value.decode()
    # the same frames through the C ABI + ctypes, and through the oracle
    arrays = rt.SceneArrays.from_scene(rt.Scene.from_name(name, DATA))
    tr = rt.RayTracer(0, w, h)
    tr.load_scene(arrays)
    tr.reset_timing()
    ref = np.zeros((h, w, 4), np.float32)
    for f in range(frames):
        p = rt.make_params(w, h, nb, spp, skybox=1, frames=f)
        tr.render(p)
        ref, _ = oracle.render(p, arrays, image=ref)
    assert np.array_equal(bits(out), bits(tr.read_image(w, h)))
    assert np.array_equal(bits(out), bits(ref))
    assert seg.value == tr.stats().segments
    tr.close()
