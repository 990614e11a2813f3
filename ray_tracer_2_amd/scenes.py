"""Scene builders for the BASELINE configs that go beyond `Scene.from_name`.

Raw (pre-BVH) mesh fixtures let the GPU box, where the reference's assets do
not exist, rebuild scenes through the same C++ pipeline (BVH build included).
"""
import os

import numpy as np

from . import _abi as A
from .scene import Scene, material, transform


def save_raw_meshes(path, scene):
    d = {}
    for i, (label, v, idx, t, m) in enumerate(scene.raw_meshes()):
        d[f"label_{i}"] = np.frombuffer(label.encode(), np.uint8)
        d[f"v_{i}"], d[f"i_{i}"] = v, idx
        d[f"t_{i}"] = np.frombuffer(bytes(t), np.uint8)
        d[f"m_{i}"] = np.frombuffer(bytes(m), np.uint8)
    np.savez_compressed(path, **d)


def load_raw_meshes(path):
    z = np.load(path)
    out, i = [], 0
    while f"v_{i}" in z:
        out.append((bytes(z[f"label_{i}"]).decode(), z[f"v_{i}"], z[f"i_{i}"],
                    A.Transform.from_buffer_copy(z[f"t_{i}"].tobytes()),
                    A.Material.from_buffer_copy(z[f"m_{i}"].tobytes())))
        i += 1
    return out


def cornell_from_raw(raw):
    """CornellBox-Original rebuilt from raw mesh fixtures (camera of Scene::cornell_box, scene.rs:914-917)."""
    sc = Scene()
    sc.set_camera((0, 1, 2), (0, 1, 0))
    for _label, v, idx, t, m in raw:
        sc.add_mesh_from_data(v, idx, xform=t, mat=m)
    return sc


# room_2's dragon material (scene.rs:602-605)
DRAGON_MATERIAL = dict(color=(0.96078, 0.11372, 0.4039, 1.0), emission_color=(1, 1, 1, 1),
                       specular_color=(1, 1, 1, 1), smoothness=0.8, specular=0.015, ior=0.0)


def cornell_dragon(cornell_raw, dragon_raw, subdivide=3):
    """BASELINE config 3 stand-in: the missing Dragon_80K.obj is replaced by
    assets/dragon.obj (8,712 triangles) with every triangle split n x n (n = 3 -> 78,408
    triangles), inside the Cornell box, with room_2's dragon material."""
    dragon = Scene()
    for _label, v, idx, _t, _m in dragon_raw:
        h = float(np.sin(-1.5708 / 2)), float(np.cos(-1.5708 / 2))
        dragon.add_mesh_from_data(v, idx, xform=transform(pos=(0.05, 1.05, 0.15), rot=(0, h[0], 0, h[1]), scale=(0.9, 0.9, 0.9)),
                                  mat=material(**DRAGON_MATERIAL))
    dragon.subdivide_meshes(subdivide)
    sc = cornell_from_raw(cornell_raw)
    for _label, v, idx, t, m in dragon.raw_meshes():
        sc.add_mesh_from_data(v, idx, xform=t, mat=m)
    sc.build()
    return sc
