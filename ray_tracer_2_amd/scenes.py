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


def cornell_dragon(cornell_raw, dragon_raw, subdivide=3, device=None):
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
    sc.build(device=device)   # device: SAH searches of the big mesh on that GPU (same result)
    return sc


def _box_mesh(lo, hi, detail=1, rng=None):
    """12 triangles, outward normals, per-face uv in [0, 1]; vertices8 + indices.  detail = k > 1: every face is a
    k x k grid of quads (12 k^2 triangles) whose inner vertices are pushed in and out along the face normal (a
    carved surface, so that the mesh's BVH is a real tree and not six coplanar sheets)."""
    if detail > 1:
        return _carved_box_mesh(lo, hi, detail, rng)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    faces = [  # (origin, du, dv, normal)
        ((lo[0], lo[1], hi[2]), (hi[0] - lo[0], 0, 0), (0, hi[1] - lo[1], 0), (0, 0, 1)),
        ((hi[0], lo[1], lo[2]), (lo[0] - hi[0], 0, 0), (0, hi[1] - lo[1], 0), (0, 0, -1)),
        ((hi[0], lo[1], hi[2]), (0, 0, lo[2] - hi[2]), (0, hi[1] - lo[1], 0), (1, 0, 0)),
        ((lo[0], lo[1], lo[2]), (0, 0, hi[2] - lo[2]), (0, hi[1] - lo[1], 0), (-1, 0, 0)),
        ((lo[0], hi[1], hi[2]), (hi[0] - lo[0], 0, 0), (0, 0, lo[2] - hi[2]), (0, 1, 0)),
        ((lo[0], lo[1], lo[2]), (hi[0] - lo[0], 0, 0), (0, 0, hi[2] - lo[2]), (0, -1, 0)),
    ]
    v, idx = [], []
    for o, du, dv, n in faces:
        o, du, dv = np.float32(o), np.float32(du), np.float32(dv)
        base = len(v)
        for (a, b) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            p = o + du * a + dv * b
            v.append([p[0], p[1], p[2], n[0], n[1], n[2], 2.0 * a, 2.0 * b])   # uv beyond 1: repeat addressing
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(v, np.float32), np.array(idx, np.uint32)


def _carved_box_mesh(lo, hi, k, rng):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    ext = hi - lo
    faces = [((lo[0], lo[1], hi[2]), (ext[0], 0, 0), (0, ext[1], 0), (0, 0, 1)),
             ((hi[0], lo[1], lo[2]), (-ext[0], 0, 0), (0, ext[1], 0), (0, 0, -1)),
             ((hi[0], lo[1], hi[2]), (0, 0, -ext[2]), (0, ext[1], 0), (1, 0, 0)),
             ((lo[0], lo[1], lo[2]), (0, 0, ext[2]), (0, ext[1], 0), (-1, 0, 0)),
             ((lo[0], hi[1], hi[2]), (ext[0], 0, 0), (0, 0, -ext[2]), (0, 1, 0)),
             ((lo[0], lo[1], lo[2]), (ext[0], 0, 0), (0, 0, ext[2]), (0, -1, 0))]
    g = np.arange(k + 1, dtype=np.float32) / np.float32(k)
    aa, bb = np.meshgrid(g, g, indexing="xy")
    inner = ((aa > 0) & (aa < 1) & (bb > 0) & (bb < 1)).astype(np.float32)
    depth = np.float32(0.04 * float(min(ext[ext > 0].min(), 20.0)))
    vs, ids = [], []
    q = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).reshape(-1).astype(np.uint32)
    quad = np.stack([q, q + 1, q + k + 2, q, q + k + 2, q + k + 1], 1).reshape(-1)
    for o, du, dv, n in faces:
        o, du, dv, n = (np.asarray(x, np.float32) for x in (o, du, dv, n))
        bump = (rng.uniform(-1.0, 1.0, aa.shape).astype(np.float32) * inner * depth)[..., None]
        pos = o + aa[..., None] * du + bb[..., None] * dv + bump * n
        v = np.zeros(((k + 1) * (k + 1), 8), np.float32)
        v[:, 0:3] = pos.reshape(-1, 3)
        v[:, 3:6] = n
        v[:, 6] = 2.0 * aa.reshape(-1)
        v[:, 7] = 2.0 * bb.reshape(-1)
        ids.append(quad + np.uint32(sum(len(x) for x in vs)))
        vs.append(v)
    return np.concatenate(vs), np.concatenate(ids).astype(np.uint32)


def sponza_standin(n_meshes=200, seed=11, detail=1):
    """BASELINE config 4 stand-in (sponza.obj and 4 of its textures are missing from the checkout):
    a many-mesh textured atrium with the structure of Scene::sponza (scene.rs:864-910) -- one OBJ's
    worth of groups sharing a single transform of scale 0.05, textured materials, plus the emissive
    quad and the emissive sphere -- built procedurally: a floor, a ceiling band, and rows of columns
    and blocks (12 triangles each), `n_meshes` meshes in all, 8 procedural textures.  `detail` = k > 1 carves every
    face into a k x k grid (12 k^2 triangles per mesh): `sponza_standin(340, detail=8)` has sponza.obj's size
    (261 k triangles in some hundreds of groups) and no longer fits the LDS."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    sc.set_camera((0, 4, 0), (0, 4, 1))   # scene.rs:867-870
    tex_ids = []
    for t in range(8):
        yy, xx = np.mgrid[0:64, 0:64]
        base = rng.randint(40, 215, 3)
        pat = ((xx // (4 + t)) + (yy // (3 + t))) % 2
        img = np.zeros((64, 64, 4), np.uint8)
        for c in range(3):
            img[..., c] = np.clip(base[c] + pat * 40 - 20 + rng.randint(-10, 10, (64, 64)), 0, 255)
        img[..., 3] = 255
        tex_ids.append(sc.add_texture_rgba8(img))
    xf = transform(scale=(0.05, 0.05, 0.05))   # scene.rs:873-877
    def textured(i):
        return material(color=(0.7, 0.7, 0.7, 1), specular_color=(1, 1, 1, 1), smoothness=0.3, specular=0.04,
                        flag=A.MATERIAL_TEXTURE, diffuse_index=tex_ids[i % len(tex_ids)])
    boxes = [((-300, -2, -150), (300, 0, 150)), ((-300, 250, -150), (300, 252, -60)), ((-300, 250, 60), (300, 252, 150))]
    k = 0
    while len(boxes) < n_meshes:
        row = k % 4
        x = -280 + (k // 4) * 23.0
        zc = (-120, -45, 45, 120)[row]
        w, d, hgt = rng.uniform(6, 10), rng.uniform(6, 10), rng.uniform(60, 240)
        boxes.append(((x - w, 0, zc - d), (x + w, hgt, zc + d)))
        k += 1
    for i, (lo, hi) in enumerate(boxes[:n_meshes]):
        v, idx = _box_mesh(lo, hi, detail, rng)
        sc.add_mesh_from_data(v, idx, xform=xf, mat=textured(i))
    h = float(np.sin(np.pi / 4)), float(np.cos(np.pi / 4))
    quad = np.array([[-1, -1, 0, 0, 0, 1, 0, 0], [1, -1, 0, 0, 0, 1, 1, 0], [1, 1, 0, 0, 0, 1, 1, 1], [-1, 1, 0, 0, 0, 1, 0, 1]], np.float32)
    sc.add_mesh_from_data(quad, [0, 1, 2, 0, 2, 3], xform=transform(pos=(-15, 60, 0), rot=(h[0], 0, 0, h[1]), scale=(40, 20, 1)),
                          mat=material(color=(0.7, 0.7, 0.7, 1), emission_color=(1, 1, 1, 1), specular_color=(1, 1, 1, 1),
                                       emission_strength=4.0, smoothness=1.0))   # scene.rs:884-892
    sc.add_sphere((5, 2, 0), 2.0, material(color=(1, 1, 1, 1), emission_color=(1, 1, 1, 1), specular_color=(1, 1, 1, 1),
                                          emission_strength=10.0, smoothness=0.0, specular=0.0))   # scene.rs:894-908
    sc.build()
    return sc
