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


# ---- config 4 stand-in with the reference's real materials and textures (round 5) ---------------------------------------
def _grid_sheet(o, du, dv, nu, nv, bump=0.0, rng=None, wave=None):
    """nu x nv quads spanning o + a du + b dv; optional relief along the sheet's normal (random bumps, or a sine wave along u:
    drapes).  Returns positions [n, 3], uvs [n, 2], faces [m, 3]."""
    o, du, dv = (np.asarray(x, np.float64) for x in (o, du, dv))
    n = np.cross(du, dv)
    n = n / np.linalg.norm(n)
    a, b = np.meshgrid(np.arange(nu + 1) / nu, np.arange(nv + 1) / nv, indexing="xy")
    pos = o + a[..., None] * du + b[..., None] * dv
    if wave is not None:
        pos = pos + (np.sin(a * wave[0] * 2 * np.pi) * wave[1])[..., None] * n
    if bump and rng is not None:
        inner = ((a > 0) & (a < 1) & (b > 0) & (b < 1)).astype(np.float64)
        pos = pos + (rng.uniform(-1, 1, a.shape) * inner * bump)[..., None] * n
    uv = np.stack([a * max(1, nu // 4), b * max(1, nv // 4)], -1)
    q = (np.arange(nv)[:, None] * (nu + 1) + np.arange(nu)[None, :]).reshape(-1)
    f = np.stack([q, q + 1, q + nu + 2, q, q + nu + 2, q + nu + 1], 1).reshape(-1, 3)
    return pos.reshape(-1, 3), uv.reshape(-1, 2), f


def _cylinder(c, r, h, seg, rings, taper=0.0):
    """A column: seg x rings quads around the y axis from c (base centre), radius r (tapering by `taper`), height h; open ends."""
    th, yy = np.meshgrid(np.arange(seg + 1) / seg * 2 * np.pi, np.arange(rings + 1) / rings, indexing="xy")
    rr = r * (1.0 - taper * yy) * (1.0 + 0.06 * np.sin(yy * 9.0))
    pos = np.stack([c[0] + rr * np.cos(th), c[1] + yy * h, c[2] + rr * np.sin(th)], -1)
    uv = np.stack([th / (2 * np.pi) * 2.0, yy * 4.0], -1)
    q = (np.arange(rings)[:, None] * (seg + 1) + np.arange(seg)[None, :]).reshape(-1)
    f = np.stack([q, q + seg + 2, q + 1, q, q + seg + 1, q + seg + 2], 1).reshape(-1, 3)
    return pos.reshape(-1, 3), uv.reshape(-1, 2), f


def _blob(c, r, seg, rings, squash=1.0):
    """A vase / lion head: a lat-long sphere of radius r around c, squashed in y."""
    th, ph = np.meshgrid(np.arange(seg + 1) / seg * 2 * np.pi, (np.arange(rings + 1) / rings) * np.pi, indexing="xy")
    rr = r * (1.0 + 0.15 * np.sin(3 * ph))
    pos = np.stack([c[0] + rr * np.sin(ph) * np.cos(th), c[1] - rr * np.cos(ph) * squash, c[2] + rr * np.sin(ph) * np.sin(th)], -1)
    uv = np.stack([th / (2 * np.pi), ph / np.pi], -1)
    q = (np.arange(rings)[:, None] * (seg + 1) + np.arange(seg)[None, :]).reshape(-1)
    f = np.stack([q, q + seg + 2, q + 1, q, q + seg + 1, q + seg + 2], 1).reshape(-1, 3)
    return pos.reshape(-1, 3), uv.reshape(-1, 2), f


def _write_obj(path, mtl_name, groups, with_normals):
    """groups: [(name, material, positions, uvs, faces)].  One `g` + `usemtl` per group (tobj 4.0.3 starts a model at each);
    `vn` lines only when asked -- without them the loader synthesises smooth normals (src/core/asset.rs:224-261)."""
    out = [f"mtllib {mtl_name}"]
    base = 1
    for name, mat, pos, uv, faces in groups:
        out.append(f"g {name}")
        out.append(f"usemtl {mat}")
        out += [f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}" for p in pos]
        out += [f"vt {t[0]:.6f} {t[1]:.6f}" for t in uv]
        if with_normals:
            nrm = np.zeros_like(pos)
            e1, e2 = pos[faces[:, 1]] - pos[faces[:, 0]], pos[faces[:, 2]] - pos[faces[:, 1]]
            fn = np.cross(e1, e2)
            for k in range(3):
                np.add.at(nrm, faces[:, k], fn)
            nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
            out += [f"vn {n[0]:.6f} {n[1]:.6f} {n[2]:.6f}" for n in nrm]
            out += [f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}" for a, b, c in faces + base]
        else:
            out += [f"f {a}/{a} {b}/{b} {c}/{c}" for a, b, c in faces + base]
        base += len(pos)
    open(path, "w").write("\n".join(out) + "\n")


SPONZA_FIXTURES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sponza")


def sponza_hetero(seed=5, fixtures=SPONZA_FIXTURES, workdir=None):
    """BASELINE config 4 stand-in built from what the reference DOES hold of sponza (round 5): the 25
    materials of assets/sponza.mtl and 25 of its textures at native resolution (tests/golden/sponza/: data fixtures), loaded
    through the OBJ / MTL / PNG loader exactly as Scene::sponza would load sponza.obj (src/scene/scene.rs:864-910,
    src/core/asset.rs:102-330), on PROCEDURAL geometry with sponza.obj's shape of statistics: about 390 groups of 12 to
    about 40,000 triangles (two orders of magnitude, a few big ones holding most of the triangles), about 262 k in all --
    flat and relief walls and floors, columns, arches, vases, lion heads, drapes.  Unlike Scene::sponza (ONE transform for the
    whole OBJ) the groups come under FIVE transforms -- different scales, rotations, a non-uniform one -- so that runs of
    meshes sharing a local space, top-level trees and ITEM_PRUNE are not universal.  Two OBJ files carry `vn`, three leave the
    normals to the loader.  Plus Scene::sponza's emissive quad and sphere (scene.rs:879-908)."""
    import tempfile
    rng = np.random.RandomState(seed)
    work = workdir or tempfile.mkdtemp(prefix="rt2_sponza_")
    for name in ("sponza_standin.mtl", "textures"):
        dst = os.path.join(work, name)
        if not os.path.exists(dst):
            os.symlink(os.path.join(fixtures, name), dst)
    G = {k: [] for k in "ABCDE"}

    def add(obj, name, mat, shape):
        G[obj].append((f"{name}_{len(G[obj])}", mat, *shape))

    # A: the hall (the reference's transform: scale 0.05) -- floor tiles, four relief walls (the big meshes), ceiling, columns
    for i in range(6):
        for j in range(3):
            add("A", "floor", "floor", _grid_sheet((-300 + i * 100, 0, -150 + j * 100), (0, 0, 100), (100, 0, 0), 2 + (i + j) % 3, 2 + (i + j) % 3))   # (normal +y)
    walls = [((-300, 0, -150), (600, 0, 0), (0, 250, 0), 160, 125), ((300, 0, 150), (-600, 0, 0), (0, 250, 0), 150, 110),
             ((-300, 0, 150), (0, 0, -300), (0, 250, 0), 100, 90), ((300, 0, -150), (0, 0, 300), (0, 250, 0), 90, 80)]
    for k, (o, du, dv, nu, nv) in enumerate(walls):   # 40,000 / 33,000 / 18,000 / 14,400 triangles
        add("A", "wall", "bricks", _grid_sheet(o, du, dv, nu, nv, bump=1.5, rng=rng))
    for i in range(4):
        add("A", "ceiling", "ceiling", _grid_sheet((-300 + i * 150, 250, 150), (0, 0, -300), (150, 0, 0), 50, 40, bump=0.8, rng=rng))   # (normal -y)
    add("A", "roof", "roof", _grid_sheet((-300, 252, -150), (600, 0, 0), (0, 0, 300), 4, 2))
    cols = ["column_a", "column_b", "column_c"]
    for i in range(44):
        x, z = -280 + (i // 4) * 52.0, (-120, -45, 45, 120)[i % 4]
        seg, rings = int(rng.choice([8, 12, 16, 24, 32])), int(rng.choice([6, 10, 16, 24]))
        add("A", "column", cols[i % 3], _cylinder((x, 0, z), rng.uniform(5, 9), rng.uniform(120, 240), seg, rings, taper=0.15))
    for i in range(30):
        add("A", "detail", "details", _grid_sheet((-290 + i * 19.0, rng.uniform(20, 200), -149.0), (14, 0, 0), (0, 10, 0), 1 + i % 3, 1 + i % 2))
    for i in range(12):   # arches between the columns: bent sheets
        add("A", "arch", "arch", _grid_sheet((-280 + i * 48.0, 200, -120), (40, 0, 0), (0, 0, 75), 40, 20, wave=(0.5, 18.0)))
    # B: the upper gallery (same scale, lifted) -- smaller columns, arches, details, flagpoles
    for i in range(40):
        x, z = -270 + (i // 2) * 28.0, (-130, 130)[i % 2]
        add("B", "gcolumn", cols[(i + 1) % 3], _cylinder((x, 0, z), rng.uniform(3, 5), rng.uniform(50, 80), int(rng.choice([6, 8, 12])), int(rng.choice([3, 4, 8])), taper=0.1))
    for i in range(20):
        add("B", "garch", "arch", _grid_sheet((-270 + i * 27.0, 70, (-130, 118)[i % 2]), (24, 0, 0), (0, 0, 12), 8, 2, wave=(0.5, 6.0)))
    for i in range(16):
        add("B", "flagpole", "flagpole", _cylinder((-250 + i * 33.0, 10, (-100, 100)[i % 2]), 0.8, 60, 6, 2))
    # C: props (another scale, rotated) -- vases, plants, lion heads, chains: many small meshes, a few detailed ones
    for i in range(60):
        mat = ["vase", "vase_round", "vase_hanging", "Material__57", "Material__25", "chain"][i % 6]
        seg, rings = ((6, 4), (8, 6), (12, 8), (16, 12), (24, 16), (48, 32))[int(rng.choice(6, p=[0.3, 0.25, 0.2, 0.15, 0.07, 0.03]))]
        add("C", "prop", mat, _blob((rng.uniform(-200, 200), rng.uniform(8, 60), rng.uniform(-90, 90)), rng.uniform(4, 12), seg, rings, squash=rng.uniform(0.8, 1.6)))
    for i in range(50):
        add("C", "chip", ["Material__47", "Material__298", "leaf"][i % 3], _grid_sheet((rng.uniform(-220, 220), rng.uniform(1, 90), rng.uniform(-100, 100)), (rng.uniform(3, 9), 0, 0), (0, rng.uniform(3, 9), 0), 1, 1 + i % 2))
    # D: drapes (a non-uniform scale, rotated) -- wavy sheets of all seven fabric materials
    fabrics = ["fabric_a", "fabric_c", "fabric_d", "fabric_e", "fabric_f", "fabric_g"]
    for i in range(36):
        nu, nv = ((10, 6), (20, 12), (40, 20), (60, 30))[int(rng.choice(4, p=[0.4, 0.3, 0.2, 0.1]))]
        add("D", "drape", fabrics[i % 6], _grid_sheet((-260 + (i % 18) * 29.0, 60 + 70 * (i // 18), (-100, 100)[i % 2]), (26, 0, 0), (0, 60, 0), nu, nv, wave=(3.0, 2.5)))
    # E: the far wing (rotated by a quarter turn, moved) -- more walls, columns, details
    for i in range(3):
        add("E", "wwall", "bricks", _grid_sheet((-150, 0, -100 + i * 100), (300, 0, 0), (0, 200, 0), 96 - 20 * i, 74 - 12 * i, bump=1.2, rng=rng))
    for i in range(24):
        add("E", "wcolumn", cols[i % 3], _cylinder((-140 + (i // 2) * 25.0, 0, (-60, 60)[i % 2]), rng.uniform(4, 7), rng.uniform(100, 180), int(rng.choice([8, 12, 16])), int(rng.choice([4, 8, 12])), taper=0.12))
    for i in range(30):
        add("E", "wdetail", ["details", "Material__298", "roof"][i % 3], _grid_sheet((-145 + i * 9.5, rng.uniform(10, 150), -99.0), (7, 0, 0), (0, 7, 0), 1, 1))
    h2 = lambda a: (float(np.sin(a / 2)), float(np.cos(a / 2)))
    xforms = {"A": transform(scale=(0.05, 0.05, 0.05)),                                                    # scene.rs:873-877
              "B": transform(pos=(0.0, 6.5, 0.0), scale=(0.05, 0.05, 0.05)),
              "C": transform(pos=(2.0, 0.0, 1.0), rot=(0, h2(0.6)[0], 0, h2(0.6)[1]), scale=(0.03, 0.03, 0.03)),
              "D": transform(pos=(-1.0, 0.0, 0.5), rot=(0, h2(-0.4)[0], 0, h2(-0.4)[1]), scale=(0.05, 0.06, 0.05)),
              "E": transform(pos=(0.0, 0.0, -9.0), rot=(0, h2(np.pi / 2)[0], 0, h2(np.pi / 2)[1]), scale=(0.05, 0.05, 0.05))}
    sc = Scene()
    sc.set_camera((0, 4, 0), (0, 4, 1))   # scene.rs:867-870
    for key in "ABCDE":
        _write_obj(os.path.join(work, f"hetero_{key}.obj"), "sponza_standin.mtl", G[key], with_normals=key in "AD")
        sc.add_mesh_from_file(f"hetero_{key}.obj", xform=xforms[key], use_mtl=True, assets_dir=work)
    hq = h2(np.pi / 2)
    quad = np.array([[-1, -1, 0, 0, 0, 1, 0, 0], [1, -1, 0, 0, 0, 1, 1, 0], [1, 1, 0, 0, 0, 1, 1, 1], [-1, 1, 0, 0, 0, 1, 0, 1]], np.float32)
    sc.add_mesh_from_data(quad, [0, 1, 2, 0, 2, 3], xform=transform(pos=(-15, 60, 0), rot=(hq[0], 0, 0, hq[1]), scale=(40, 20, 1)),
                          mat=material(color=(0.7, 0.7, 0.7, 1), emission_color=(1, 1, 1, 1), specular_color=(1, 1, 1, 1),
                                       emission_strength=4.0, smoothness=1.0))   # scene.rs:884-892
    sc.add_sphere((5, 2, 0), 2.0, material(color=(1, 1, 1, 1), emission_color=(1, 1, 1, 1), specular_color=(1, 1, 1, 1),
                                          emission_strength=10.0, smoothness=0.0, specular=0.0))   # scene.rs:894-908
    sc.build()
    return sc
