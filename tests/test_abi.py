"""The C-ABI library loads without a GPU and exports every symbol that
include/rt_abi.h declares; struct layouts match the reference's byte layout."""
import ctypes as C
import os
import re

from conftest import ROOT


def header_functions(header="rt_abi.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("rt_"))


def test_every_declared_symbol_is_exported(rt):
    L = rt.load()
    declared = header_functions()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in rt_abi.h but not exported"
    from ray_tracer_2_amd.lib import EXPORTS
    assert sorted(EXPORTS) == declared


def test_the_product_library_exports_no_test_entry_points(rt):
    """Round 5: the rt_test_* entry points live in include/rt_test_abi.h and in
    librt2_mi355x_test.so (the same sources + -DRT_TEST_ENTRIES=1); the product exports include/rt_abi.h and nothing else."""
    from ray_tracer_2_amd import lib
    product = exported_symbols(lib.LIB_PATH)
    assert not [s for s in product if s.startswith("rt_test_")], product
    assert not [s for s in header_functions() if s.startswith("rt_test_")]
    declared = header_functions("rt_test_abi.h")
    assert sorted(lib.TEST_EXPORTS) == declared and all(s.startswith("rt_test_") for s in declared)
    T = rt.load_test()
    for name in declared + header_functions():
        assert hasattr(T, name), f"{name} missing from the test library"
    # (rt_diag_* / rt_wave_times only exist in the diagnostic builds)
    assert [s for s in product if not s.startswith("rt_scene_") and s not in header_functions()] == []


def test_prototypes_take_as_many_arguments_as_the_header_declares(rt):
    """ctypes checks the argument COUNT of a call against argtypes: a prototype in lib.py with one argument too many makes
    every call of that function a TypeError (rt_set_stream once: only bench.py --gpus N called it)."""
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read() + open(os.path.join(ROOT, "include", "rt_test_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    L = rt.load_test()   # (every prototype of both headers; the product's bindings are the same table minus rt_test_*)
    seen = 0
    for name, args in re.findall(r"\b(rt_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = args.strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        fn = getattr(L, name)
        assert fn.argtypes is not None and len(fn.argtypes) == n, (name, n, fn.argtypes)
        seen += 1
    assert seen >= 40


def test_struct_sizes_match_reference_layout(rt):
    # Params 48, Material 96, Sphere 112, MeshUniform 240,
    # Node 48, PackedTriangle 96, CameraUniform 84, SceneUniform 128
    L = rt.load()
    sizes = (C.c_uint32 * 8)()
    L.rt_abi_sizes(C.byref(sizes))
    assert list(sizes) == [48, 96, 112, 240, 48, 96, 84, 128]


def test_field_offsets(rt):
    from ray_tracer_2_amd import _abi as A
    assert A.Params.frames.offset == 20 and A.Params.debug_scale.offset == 32
    assert A.Material.absorption_strength.offset == 64 and A.Material.flag.offset == 84
    assert A.MeshUniform.node_offset.offset == 128 and A.MeshUniform.material.offset == 144
    assert A.Node.aabb_min.offset == 16 and A.Node.aabb_max.offset == 32
    assert A.PackedTriangle.n1.offset == 48 and A.PackedTriangle.uv31.offset == 92
    assert A.SceneUniform.camera.offset == 16 and A.SceneUniform.nodes.offset == 100
    assert A.CameraUniform.view_params.offset == 64


def test_no_device_is_a_clean_error(rt):
    """Without a GPU rt_create must fail loudly (no CPU fallback), not crash."""
    L = rt.load()
    if L.rt_device_count() > 0:
        return
    h = C.c_void_p()
    rc = L.rt_create(0, 64, 64, C.byref(h))
    assert rc == -3 and not h
    assert b"no HIP device" in L.rt_last_error(None)


def test_version_string(rt):
    assert b"gfx950" in rt.load().rt_version()
