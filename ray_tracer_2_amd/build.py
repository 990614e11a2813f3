"""Build recipes: the HIP/C++ product library and (separately) the CPU oracle.

`hipcc` cross-compiles gfx950 without a GPU.  The product library is built
in-tree (ray_tracer_2_amd/librt2_mi355x.so) so it travels with the repo
snapshot; the oracle goes to oracle/_build/.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ray_tracer_2_amd", "csrc")
PRODUCT_SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "librt_oracle.so")

PRODUCT_SOURCES = [
    "rt_kernel.hip", "rt_api.hip", "rt_bvh_search.hip", "host/obj_loader.cpp", "host/bvh.cpp", "host/scene.cpp",
    "host/png_decode.cpp", "host/scene_capi.cpp", "host/ray_tracer.cpp",
]
PRODUCT_HEADERS = [
    "rt_transc.h", "rt_texture.h", "rt_srgb_lut.h", "rt_device.h", "rt_rccl.h", "host/glam_math.h",
    "host/obj_loader.h", "host/bvh.h", "host/scene.h", "host/ray_tracer.hpp",
    "../../include/rt_abi.h", "../../include/rt_test_abi.h", "experiments/rt_wavefront.inl", "experiments/rt_wavefront_launch.inl", "experiments/rt_api_wavefront.inl", "experiments/rt_api_hybrid_blob.inl",
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def hipcc_path():
    for p in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(p) or p == "hipcc":
            return p


def build_product(force=False, extra_flags=(), out=None, jobs=4):
    """Compile every product source to an object (in parallel, only the ones whose inputs changed)
    and link librt2_mi355x.so.  `extra_flags` / `out` build an instrumented variant beside it
    (tools/diag.py)."""
    import concurrent.futures
    import hashlib
    out = out or PRODUCT_SO
    srcs = [os.path.join(CSRC, s) for s in PRODUCT_SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in PRODUCT_HEADERS] + [os.path.abspath(__file__)]   # (flags live in this file)
    # -ffp-contract=off: the canonical arithmetic is unfused (every fused
    # operation is an explicit fma in rt_transc.h); division and sqrt stay
    # correctly rounded (hipcc default, no fast-math).
    # -fno-slp-vectorize: the SLP vectoriser's automatic v_pk_*_f32 packing pays for itself in
    # v_mov shuffles here (measured: 2.22 -> 2.09 ms on config 2 without it).
    flags = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
             "-fno-slp-vectorize", "-fPIC", "-Wall", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"),
             *extra_flags]
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "product"
    objdir = os.path.join(ROOT, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    objs, todo = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            todo.append([hipcc_path(), *flags, "-c", s, "-o", o])
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(_run, todo))
    if todo or _newer(out, objs):
        _run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-ldl", "-o", out])
    return out


TEST_SO = os.path.join(ROOT, "ray_tracer_2_amd", "librt2_mi355x_test.so")


def build_test_library(force=False):
    """librt2_mi355x_test.so: the product's sources and flags plus -DRT_TEST_ENTRIES=1 -- the test-only entry points of
    include/rt_test_abi.h, which the product library does not export (round 5)."""
    return build_product(force=force, extra_flags=("-DRT_TEST_ENTRIES=1",), out=TEST_SO)


CLASS_DRIVER_SO = os.path.join(ROOT, "tests", "_build", "librt2_class_driver.so")


def build_class_driver(force=False):
    """tests/cpp/ray_tracer_class_driver.cpp: the C++ test driver of rt2::RayTracer / rt2::FrameParams,
    linked against the product library (tests/test_gpu_cpp_class.py)."""
    src = os.path.join(ROOT, "tests", "cpp", "ray_tracer_class_driver.cpp")
    deps = [src, PRODUCT_SO] + [os.path.join(CSRC, h) for h in ("host/ray_tracer.hpp", "host/scene.h", "host/bvh.h")]
    if not force and not _newer(CLASS_DRIVER_SO, deps):
        return CLASS_DRIVER_SO
    os.makedirs(os.path.dirname(CLASS_DRIVER_SO), exist_ok=True)
    _run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-I", os.path.join(ROOT, "include"), src,
          "-L", os.path.dirname(PRODUCT_SO), "-lrt2_mi355x", "-Wl,-rpath,$ORIGIN/../../ray_tracer_2_amd", "-o", CLASS_DRIVER_SO])
    return CLASS_DRIVER_SO


FAKE_RCCL_SO = os.path.join(ROOT, "tests", "_build", "libfake_rccl.so")


def build_fake_rccl(force=False):
    """tests/cpp/fake_rccl.c: the RCCL-shaped stub that fails on request (tests/test_rccl_group.py)."""
    src = os.path.join(ROOT, "tests", "cpp", "fake_rccl.c")
    if not force and not _newer(FAKE_RCCL_SO, [src]):
        return FAKE_RCCL_SO
    os.makedirs(os.path.dirname(FAKE_RCCL_SO), exist_ok=True)
    _run(["gcc", "-std=c11", "-O1", "-fPIC", "-shared", "-Wall", "-Wextra", src, "-o", FAKE_RCCL_SO])
    return FAKE_RCCL_SO


HOST_SANITIZER_DRIVER = os.path.join(ROOT, "tests", "_build", "host_sanitizer_driver")


def build_host_sanitizer_driver(force=False):
    """tests/cpp/host_sanitizer_driver.cpp + the host loaders (OBJ / MTL / PNG / scene / BVH, no HIP) with
    AddressSanitizer and UndefinedBehaviorSanitizer, CPU only (tests/test_host_malformed_inputs.py)."""
    host = [os.path.join(CSRC, "host", f) for f in ("obj_loader.cpp", "bvh.cpp", "scene.cpp", "png_decode.cpp", "scene_capi.cpp")]
    src = os.path.join(ROOT, "tests", "cpp", "host_sanitizer_driver.cpp")
    deps = [src, *host] + [os.path.join(CSRC, h) for h in PRODUCT_HEADERS if h.startswith("host/")] + [os.path.join(ROOT, "include", "rt_abi.h")]
    if not force and not _newer(HOST_SANITIZER_DRIVER, deps):
        return HOST_SANITIZER_DRIVER
    os.makedirs(os.path.dirname(HOST_SANITIZER_DRIVER), exist_ok=True)
    _run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
          "-Wall", "-I", os.path.join(ROOT, "include"), src, *host, "-lz", "-o", HOST_SANITIZER_DRIVER])
    return HOST_SANITIZER_DRIVER


def source_hash():
    """Identity of the render kernels a profile was taken on: hash of the device sources and of this
    file (the compile flags).  profiles/latest_traffic.json carries it; bench.py only reports counter
    figures measured on the build it is running."""
    import hashlib
    hh = hashlib.sha1()
    for f in ("rt_kernel.hip", "rt_api.hip", "rt_device.h", "rt_transc.h", "rt_texture.h"):
        hh.update(open(os.path.join(CSRC, f), "rb").read())
    hh.update(open(os.path.abspath(__file__), "rb").read())
    return hh.hexdigest()[:16]


def build_oracle(force=False):
    src = os.path.join(ROOT, "oracle", "shader_oracle.cpp")
    deps = [src] + [os.path.join(CSRC, h) for h in ("rt_transc.h", "rt_texture.h", "rt_srgb_lut.h")]
    deps.append(os.path.join(ROOT, "include", "rt_abi.h"))
    if not force and not _newer(ORACLE_SO, deps):
        return ORACLE_SO
    os.makedirs(os.path.dirname(ORACLE_SO), exist_ok=True)
    flags = ["-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
             "-pthread", "-Wall", "-Wextra"]
    try:
        if " fma " in open("/proc/cpuinfo").read():
            flags.append("-mfma")  # __builtin_fmaf -> vfmadd (same result as libm fmaf)
    except OSError:
        pass
    _run(["g++", *flags, src, "-o", ORACLE_SO])
    return ORACLE_SO


if __name__ == "__main__":
    build_product(force="--force" in sys.argv)
    build_oracle(force="--force" in sys.argv)
