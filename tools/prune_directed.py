#!/usr/bin/env python3
"""Directed scenes for the hypothesis behind cross-mesh pruning.

Random scenes only sample AROUND the geometry that could break "a triangle hit at parameter t is not found under a box
whose entry distance is far beyond t".  These families AIM at it: every primary ray of a 1920x1080 frame is a directed ray
(the camera is a pencil a few 1e-4 rad wide, so the pixel pitch is about 2e-7 rad), and every family is a many-mesh item
(>= 8 meshes with internal roots under one transform: a top-level tree, ITEM_PRUNE) with near occluders in front of far
geometry, so that the far geometry's boxes are the ones the pruned walk refuses.

  blades    far strips whose planes pass within 0 .. 1e-4 of the camera origin, seen at 1e-7 .. 1e-4 rad (a, b): the shader's
            own t = dot(ao, n) / det is a quotient of two cancelling sums there
  faces     rays within 1e-7 .. 1e-4 rad of leaf-box faces: far tiles whose boxes are flat and axis-aligned, the view axis inside
            the plane of their faces (a)
  large     coordinates around 1e3, hits at t = 1e-4 .. 1e-2: the slab test's (bmin - o) cancels (c)
  slivers   far geometry made of needle triangles, sin(phi) about 1e-6 (d)
  ties      a near and a far surface at almost equal world distance, in different meshes; coincident double-sided sheets (e)

`families()` returns [(name, SceneArrays, camera uniform)], used by tests/test_gpu_prune_directed.py (GPU: pruned == unpruned
== oracle, bit for bit) and by this script, which runs the ORACLE's census of every triangle hit against its leaf box on the
same rays (CPU only):

    python tools/prune_directed.py [--frames 1] > profiles/r05_prune_directed_census.txt
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402
from ray_tracer_2_amd import _abi as A  # noqa: E402
from ray_tracer_2_amd.scene import Scene, material, transform  # noqa: E402

W, H = 1920, 1080


def _mat(k):
    """A distinct emissive colour per mesh: the image says which mesh a primary ray hit."""
    c = ((k * 37 % 97) / 97.0 * 0.8 + 0.1, (k * 53 % 89) / 89.0 * 0.8 + 0.1, (k * 71 % 83) / 83.0 * 0.8 + 0.1, 1.0)
    return material(color=c, emission_color=c, specular_color=(1, 1, 1, 1), emission_strength=1.0, smoothness=0.2, specular=0.1)


def _strip(p00, p01, p10, p11, n_seg, flip=False):
    """A quad (p00 -> p01 along u, p00 -> p10 along v) cut into n_seg quads along v, two triangles each."""
    p00, p01, p10, p11 = (np.asarray(p, np.float64) for p in (p00, p01, p10, p11))
    v, idx = [], []
    for s in range(n_seg + 1):
        f = s / n_seg
        a, b = p00 + (p10 - p00) * f, p01 + (p11 - p01) * f
        for p, uu in ((a, 0.0), (b, 1.0)):
            v.append([p[0], p[1], p[2], 0, 0, 0, uu, f])
    for s in range(n_seg):
        o = 2 * s
        tri = [o, o + 1, o + 3, o, o + 3, o + 2]
        if flip:
            tri = [o, o + 3, o + 1, o, o + 2, o + 3]
        idx += tri
    v = np.array(v, np.float32)
    # face normal for the shading record (any unit vector will do: the tests compare images, not beauty)
    e1, e2 = v[1, :3] - v[0, :3], v[2, :3] - v[0, :3]
    n = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    n = n / (np.linalg.norm(n) or 1.0) * (-1.0 if flip else 1.0)
    v[:, 3:6] = n.astype(np.float32)
    return v, np.array(idx, np.uint32)


def _camera(origin, pw, ph, axis="z"):
    """CameraUniform of a pencil of rays around +z (or +x) from `origin`: view_params = (pw, ph, 1), i.e. pixel (x, y)
    looks along (uv.x - 0.5) pw, (uv.y - 0.5) ph, 1 (wgsl:479-482)."""
    cam = A.CameraUniform()
    m = np.eye(4, dtype=np.float32)
    if axis == "x":   # local z -> world x, local x -> world -z, y stays
        m[:3, :3] = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32)
    m[:3, 3] = np.asarray(origin, np.float32)
    for c in range(4):
        for r in range(4):
            cam.cam_to_world[c][r] = float(m[r, c])
    cam.view_params[0], cam.view_params[1], cam.view_params[2] = pw, ph, 1.0
    cam.defocus_strength = 0.0
    cam.diverge_strength = 0.0
    return cam


def _finish(sc, cam):
    sc.set_camera((0, 0, 0), (0, 0, 1))
    sc.build()
    arrays = rt.SceneArrays.from_scene(sc)
    arrays.uniform.camera = cam
    return arrays


def _rotation(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _rotate(sc_meshes, R):
    out = []
    for v, idx, mat in sc_meshes:
        v = v.copy()
        v[:, :3] = (v[:, :3].astype(np.float64) @ R.T).astype(np.float32)
        v[:, 3:6] = (v[:, 3:6].astype(np.float64) @ R.T).astype(np.float32)
        out.append((v, idx, mat))
    return out


def blades(slivers=False, rot=None, pencil=4e-4, segments=4, betas=None, n_blades=24, n_occluders=8):
    """rot: a 3 x 3 rotation applied to the whole configuration, camera included.  Axis-aligned (rot = None) the sums
    dot(ao, n) and dot(dir, n) of wgsl:266,273 have one dominant term each and hardly cancel; in a generic orientation every
    term is of the size of |ao| |n| and the sums are what is left of them: the case to aim at."""
    meshes = []
    pw, ph = pencil, pencil * 0.5625
    k = 0
    # far strips, z = 60 .. 100, plane y = alpha z + beta: seen by the rays with theta_y just above / below alpha at an
    # angle of beta / z to the plane
    betas = betas or [0.0, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 3e-6]
    for b in range(n_blades):
        alpha = (-0.25 + (b + 0.5) * (0.5 / n_blades)) * pencil
        beta = betas[b % len(betas)]
        y0, y1 = alpha * 60.0 + beta, alpha * 100.0 + beta
        for flip in (False, True):   # double-sided: two coincident meshes with opposite winding
            if slivers:
                # needles: every triangle spans the strip's whole length and is 1e-4 wide at its broad end
                v, idx = [], []
                for s in range(8):
                    x0 = -0.03 + s * 0.0075
                    v += [[x0, y0, 60.0, 0, 1, 0, 0, 0], [x0 + 1e-4, y0, 60.0, 0, 1, 0, 1, 0], [x0 + 0.0075, y1, 100.0, 0, 1, 0, 1, 1]]
                    idx += [3 * s, 3 * s + 2, 3 * s + 1] if flip else [3 * s, 3 * s + 1, 3 * s + 2]
                v, idx = np.array(v, np.float32), np.array(idx, np.uint32)
            else:
                v, idx = _strip((-0.03, y0, 60.0), (0.03, y0, 60.0), (-0.03, y1, 100.0), (0.03, y1, 100.0), segments, flip)
            meshes.append((v, idx, _mat(k)))
            k += 1
    # near occluders: stripes across the view at depths up to just inside 60 / 1.125
    depths = [30.0, 40.0, 45.0, 50.0, 52.0, 53.0, 53.3, 48.0]
    for j, z in enumerate(depths[:n_occluders]):
        x0 = (-0.5 + j * 0.125 * 8 / n_occluders) * pencil * z
        x1 = x0 + 0.075 * 8 / n_occluders * pencil * z
        v, idx = _strip((x0, -0.5 * pencil * z, z), (x1, -0.5 * pencil * z, z), (x0, 0.5 * pencil * z, z), (x1, 0.5 * pencil * z, z), 2, flip=True)
        meshes.append((v, idx, _mat(k)))
        k += 1
    cam = _camera((0, 0, 0), pw, ph)
    if rot is not None:
        meshes = _rotate(meshes, rot)
        for c in range(3):
            for r in range(3):
                cam.cam_to_world[c][r] = float(np.float32(rot[r, c]))
    sc = Scene()
    for v, idx, mat in meshes:
        sc.add_mesh_from_data(v, idx, mat=mat)
    return _finish(sc, cam)


def faces():
    """Far axis-aligned slabs (flat boxes: their leaf boxes have faces IN the planes y = const the rays graze) behind near
    occluders; the view axis lies in the plane of a face."""
    sc = Scene()
    k = 0
    pw, ph = 4e-4, 2.25e-4
    for b in range(16):
        yb = (-1.0e-4 + (b + 0.5) * (2.0e-4 / 16)) * 80.0   # where the pencil is at z = 80
        off = [0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4][b % 8]
        # a horizontal sheet at y = off (through the camera's height up to `off`), z = 60 .. 100, drawn as 4 quads;
        # and a thin vertical fin standing on it, whose box faces are the x = const planes
        v, idx = _strip((-0.04 + 0.005 * b, off, 60.0), (-0.035 + 0.005 * b, off, 60.0), (-0.04 + 0.005 * b, off, 100.0), (-0.035 + 0.005 * b, off, 100.0), 4, flip=(b % 2 == 0))
        sc.add_mesh_from_data(v, idx, mat=_mat(k)); k += 1
        v, idx = _strip((0.0 + off, yb - 0.002, 60.0), (0.0 + off, yb + 0.002, 60.0), (0.0 + off, yb - 0.002, 100.0), (0.0 + off, yb + 0.002, 100.0), 4, flip=(b % 2 == 1))
        sc.add_mesh_from_data(v, idx, mat=_mat(k)); k += 1
    for j, z in enumerate([35.0, 44.0, 50.0, 53.0, 53.3, 47.0, 52.5, 41.0]):
        y0 = (-1.1e-4 + j * 2.8e-5) * z
        v, idx = _strip((-2e-4 * z, y0, z), (2e-4 * z, y0, z), (-2e-4 * z, y0 + 1.4e-5 * z, z), (2e-4 * z, y0 + 1.4e-5 * z, z), 2, flip=True)
        sc.add_mesh_from_data(v, idx, mat=_mat(k)); k += 1
    return _finish(sc, _camera((0, 0, 0), pw, ph))


def large():
    """Everything around (1000, 1000, 1000); a wall of tiles 1e-4 .. 1e-2 in front of the camera and more tiles behind it."""
    sc = Scene()
    k = 0
    O = np.array([1000.0, 1000.0, 1000.0])
    pw, ph = 1.0, 0.5625
    ulp = float(np.spacing(np.float32(1000.0)))
    for layer, dz in enumerate([2 * ulp, 1e-3, 1e-2, 0.1, 1.0, 10.0]):
        for j in range(4):
            # tiles of a layer cover a quarter of the view each, with gaps, so that deeper layers show through
            s = dz if dz > 1e-3 else 1e-3
            x0 = (-0.5 + j * 0.25) * pw * s
            x1 = x0 + 0.2 * pw * s
            y0, y1 = -0.3 * s, 0.3 * s
            z = dz
            v, idx = _strip(O + (x0, y0, z), O + (x1, y0, z), O + (x0, y1, z), O + (x1, y1, z), 2, flip=True)
            sc.add_mesh_from_data(v, idx, mat=_mat(k)); k += 1
    return _finish(sc, _camera(tuple(O), pw, ph))


def ties():
    """Surfaces in different meshes at almost the same world distance: parallel sheets 1 ulp .. 1e-6 (relative) apart,
    coincident double-sided sheets, behind each other, under a transform with scale 0.05 and a rotation (the pruning
    bound is derived from that matrix)."""
    sc = Scene()
    h = float(np.sin(0.3)), float(np.cos(0.3))
    xf = transform(pos=(0.01, -0.02, 0.03), rot=(0, h[0], 0, h[1]), scale=(0.05, 0.05, 0.05))
    k = 0
    pw, ph = 0.4, 0.225
    z0 = 40.0
    gaps = [0.0, float(np.spacing(np.float32(z0))), 2e-5, 1e-4, 1e-3, 1e-2, 0.1, 1.0, 5.0, 10.0]
    for j, g in enumerate(gaps):
        for part in range(2):
            # sheet j of pair `part`: the second one of a pair lies `g` behind the first and is shifted sideways by half a
            # tile, so that a ray sees the near one, the far one, or both
            x0 = -12.0 + j * 2.4 + part * 1.2
            z = z0 + g * part
            v, idx = _strip((x0, -6.0, z), (x0 + 2.0, -6.0, z), (x0, 6.0, z), (x0 + 2.0, 6.0, z), 3, flip=True)
            sc.add_mesh_from_data(v, idx, xform=xf, mat=_mat(k)); k += 1
    # a back wall of many tiles far behind (what the pruning is there to skip)
    for j in range(10):
        x0 = -14.0 + j * 2.8
        v, idx = _strip((x0, -8.0, 120.0), (x0 + 2.8, -8.0, 120.0), (x0, 8.0, 120.0), (x0 + 2.8, 8.0, 120.0), 4, flip=True)
        sc.add_mesh_from_data(v, idx, xform=xf, mat=_mat(k)); k += 1
    sc.set_camera((0, 0, 0), (0, 0, 1))
    sc.build()
    arrays = rt.SceneArrays.from_scene(sc)
    # the camera looks down the items' local +z axis: its matrix is the meshes' model_to_world rotation part
    m2w = np.array(arrays.meshes[0]["model_to_world"], np.float32).reshape(4, 4).T   # columns -> [r, c]
    cam = _camera((0, 0, 0), pw, ph)
    R = m2w[:3, :3] / np.float32(0.05)
    for c in range(3):
        for r in range(3):
            cam.cam_to_world[c][r] = float(R[r, c])
    for r in range(3):
        cam.cam_to_world[3][r] = float(m2w[r, 3])
    arrays.uniform.camera = cam
    return arrays


def families():
    R = _rotation((0.3, 0.5, 0.8), 0.9)
    wide = [1e-4, -1e-4, 3e-5, -3e-5, 3e-4, -3e-4, 1e-5, -1e-5, 1e-3, -1e-3, 3e-6, 0.0]
    return [("blades", blades()), ("blades_slivers", blades(slivers=True)),
            ("blades_rotated", blades(rot=R, pencil=4e-5, segments=32, betas=wide, n_blades=12)),
            ("blades_rotated_narrow", blades(rot=R, pencil=4e-6, segments=32, betas=[b * 0.1 for b in wide], n_blades=12)),
            ("blades_slivers_rotated", blades(slivers=True, rot=R, pencil=4e-5, betas=wide)),
            ("faces", faces()), ("large", large()), ("ties", ties()),
            # the same grazing geometry on the FEW-mesh kernels (forest and two-leaf items, the headline's kernel family): 2 strips
            # (4 meshes) and 2 occluders, axis-aligned and in the generic orientation
            ("few_blades", blades(n_blades=2, n_occluders=2)),
            ("few_blades_rotated", blades(rot=R, pencil=4e-5, segments=32, betas=wide, n_blades=2, n_occluders=2))]


def main():
    from oracle import oracle
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 1
    total_rays = 0
    for name, arrays in families():
        oracle.census(True)
        t0 = time.time()
        segs = 0
        for f in range(frames):
            _, st = oracle.render(rt.make_params(W, H, 1, 1, skybox=1, frames=f), arrays)
            segs += st.segments
        c = oracle.census(False)
        total_rays += segs
        print(f"{name}: {arrays.meshes.shape[0]} meshes, {arrays.triangles.shape[0]} triangles; {segs} rays ({W * H * frames} directed primary rays), "
              f"{int(c['hits'])} triangle hits in {time.time() - t0:.0f} s")
        print(f"   leaf-box entry beyond the hit's t: {int(c['entry_gt_t'])} ({c['entry_gt_t'] / max(c['hits'], 1):.2e} of the hits); "
              f"by > 1e-6: {int(c['gt_1e-6'])}, > 1e-4: {int(c['gt_1e-4'])}, > 1 %: {int(c['gt_1pct'])}, > 12.5 %: {int(c['gt_12.5pct'])}; "
              f"largest entry / t = {c['max_ratio']:.9g}", flush=True)
    print(f"total {total_rays} rays")


if __name__ == "__main__":
    main()
