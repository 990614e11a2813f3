"""ray_tracer_2_amd -- MI355X-native hot path of addiswebb/ray_tracer_2.

The product is the C-ABI library `librt2_mi355x.so` (include/rt_abi.h):
hand-written HIP for gfx950 plus the C++ host-side scene pipeline.  This
package is the thin Python face used by tests and bench.py.
"""
from ._abi import (CameraUniform, Material, MeshUniform, Node, PackedTriangle, Params,  # noqa: F401
                   SceneUniform, Sphere, make_params)
from .lib import LIB_PATH, RtError, load  # noqa: F401
from .ray_tracer import RayTracer, read_multi_frame, render_multi  # noqa: F401
from .scene import Scene, SceneArrays, material, transform  # noqa: F401
