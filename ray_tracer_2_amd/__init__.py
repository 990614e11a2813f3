"""ray_tracer_2_amd -- MI355X-native hot path of addiswebb/ray_tracer_2.

The product is the C-ABI library `librt2_mi355x.so` (include/rt_abi.h):
hand-written HIP for gfx950 plus the C++ host-side scene pipeline.  This
package is the thin Python face used by tests and bench.py.

In a process that also uses PyTorch on the GPU, import torch FIRST (bench.py does): the torch wheel brings its own
HIP runtime (ROCm 7.0), the library is linked against the image's (ROCm 7.2), and whichever is loaded first serves
both -- torch on the image's runtime reports "no ROCm-capable device" (tests/_bench_plumbing.py runs in a process
of its own for that reason).
"""
import os as _os

# The pipelined single frames keep up to four streams of a handle busy; ROCm maps a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise (csrc/rt_api.hip,
# rt2_request_hw_queues).  Has to be in the environment before the HIP runtime initialises; a host's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._abi import (CameraUniform, Material, MeshUniform, Node, PackedTriangle, Params,  # noqa: F401
                   SceneUniform, Sphere, make_params)
from .lib import LIB_PATH, RtError, load, load_test  # noqa: F401
from .ray_tracer import RayTracer, read_multi_frame, render_multi  # noqa: F401
from .scene import Scene, SceneArrays, material, transform  # noqa: F401
