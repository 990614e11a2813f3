"""CPU ORACLE for the host-side scene pipeline -- TEST INFRASTRUCTURE.

Independent numpy/pure-Python restatements (float32 arithmetic via numpy
scalars) of:
  * tobj 4.0.3's OBJ/MTL loading as configured by the reference
    (src/core/asset.rs:110-118; crate not vendored -> published behaviour,
    SURVEY.md 8a-6),
  * AssetManager::load_model's material mapping, normal synthesis and vertex
    unrolling (src/core/asset.rs:141-205, 224-310),
  * BVH::build / subdivide / find_best_split / evaluate_sah
    (src/core/bvh.rs:208-470).
PARITY UNPINNED (the reference has no fixtures); used to cross-check the C++
implementation in ray_tracer_2_amd/csrc/host/.
"""
import math
import os

import numpy as np

f32 = np.float32
MISSING = None


# --------------------------------------------------------------------------
# tobj-style OBJ / MTL parsing
# --------------------------------------------------------------------------
def _parse_index(tok, count):
    if tok == "":
        return MISSING
    v = int(tok)
    if v < 0:
        return count + v
    assert v != 0
    return v - 1


def parse_mtl(text):
    mats, cur = [], None
    for raw in text.splitlines():
        line = raw.strip()
        w = line.split()
        if not w or w[0].startswith("#"):
            continue
        k = w[0]
        if k == "newmtl":
            if cur is not None:
                mats.append(cur)
            cur = {"name": line[6:].strip(), "unknown": {}}
        elif k in ("Kd", "Ks"):
            cur[k] = [f32(x) for x in w[1:4]]
        elif k in ("Ns", "Ni"):
            cur[k] = f32(w[1])
        elif k == "illum":
            cur["illum"] = int(w[1])
        elif k == "map_Kd":
            cur["map_Kd"] = line[6:].strip()
        elif k in ("Ka", "d", "map_Ka", "map_Ks", "map_Ns", "map_ns", "bump", "map_bump", "map_Bump", "map_d"):
            pass
        else:
            cur["unknown"][k] = line[len(k):].strip()
    if cur is not None:
        mats.append(cur)
    return mats


def load_obj(path):
    """Returns (models, materials); a model = dict(name, material_id, faces) with
    faces as lists of (v, vt, vn) global indices, already triangulated the tobj way."""
    pos, tex, nrm = [], [], []
    models, faces = [], []
    name, mat_id = "unnamed_object", None
    materials, mat_map = [], {}
    base = os.path.dirname(path)

    def flush(nm):
        tris = []
        for f in faces:
            if len(f) == 1:
                tris.append((f[0], f[0], f[0]))
            elif len(f) == 2:
                tris.append((f[0], f[1], f[1]))
            elif len(f) == 3:
                tris.append(tuple(f))
            elif len(f) == 4:
                tris.append((f[0], f[1], f[2]))
                tris.append((f[0], f[2], f[3]))
            else:
                b = 1
                for c in range(2, len(f)):
                    tris.append((f[0], f[b], f[c]))
                    b = c
        models.append({"name": nm, "material_id": mat_id, "tris": tris})
        faces.clear()

    for raw in open(path).read().splitlines():
        w = raw.split()
        if not w or w[0] == "#":
            continue
        k = w[0]
        if k == "v":
            pos.append([f32(x) for x in w[1:4]])
        elif k == "vt":
            tex.append([f32(x) for x in w[1:3]])
        elif k == "vn":
            nrm.append([f32(x) for x in w[1:4]])
        elif k in ("f", "l"):
            f = []
            for tok in w[1:]:
                parts = (tok.split("/") + ["", ""])[:3]
                f.append((_parse_index(parts[0], len(pos)), _parse_index(parts[1], len(tex)),
                          _parse_index(parts[2], len(nrm))))
            faces.append(f)
        elif k in ("o", "g"):
            if faces:
                flush(name)
            name = raw.strip()[1:].strip() or "unnamed_object"
        elif k == "mtllib":
            lib = raw.strip().split(None, 1)[1].strip()
            mats = parse_mtl(open(os.path.join(base, lib)).read())
            off = len(materials)
            for i, m in enumerate(mats):
                mat_map[m["name"]] = off + i
            materials += mats
        elif k == "usemtl":
            new = mat_map.get(raw.strip().split(None, 1)[1].strip())
            if new != mat_id and faces:
                flush(name)
            mat_id = new
    flush(name)
    return models, materials, np.array(pos, f32).reshape(-1, 3), np.array(tex, f32).reshape(-1, 2), \
        np.array(nrm, f32).reshape(-1, 3)


def material_from_mtl(m):
    """asset.rs:141-205 -> dict of MaterialUniform fields."""
    color = m.get("Kd", [f32(0.7)] * 3)
    spec = m.get("Ks", [f32(1.0)] * 3)
    illum = m.get("illum", 0)
    flag = 1 if illum in (4, 6, 9) else 0
    if "map_Kd" in m or "map_Disp" in m["unknown"]:   # asset.rs:151-162: a diffuse or a "map_Disp" texture makes it TEXTURE
        flag = 2
    es = f32(0.0)
    ecol = [f32(0)] * 3
    if "Ke" in m["unknown"]:
        vals = []
        for tok in m["unknown"]["Ke"].split():
            try:
                vals.append(f32(tok))
            except ValueError:
                pass
        if len(vals) == 3:
            es = max(vals)
            d = f32(1.0) if es == 0 else es
            ecol = [v / d for v in vals]
    ns = m.get("Ns", f32(0.0))
    sm = np.sqrt(f32(ns / f32(100.0)))
    return {
        "color": [color[0], color[1], color[2], f32(1)],
        "emission_color": [ecol[0], ecol[1], ecol[2], f32(1)],
        "specular_color": [spec[0], spec[1], spec[2], f32(1)],
        "emission_strength": f32(es * f32(2.0)),
        "smoothness": f32(min(max(sm, f32(0)), f32(1))),
        "specular": f32(min(max(max(spec), f32(0)), f32(1))),
        "ior": m.get("Ni", f32(1.0)),
        "flag": flag,
    }


def unroll_model(model, pos, tex, nrm):
    """asset.rs:208-327 for one tobj model -> (positions, normals, uvs) per index."""
    tris = model["tris"]
    has_n = len(nrm) > 0 and all(v[2] is not None for t in tris for v in t)
    has_t = len(tex) > 0 and all(v[1] is not None for t in tris for v in t)
    P = np.array([[pos[v[0]] for v in t] for t in tris], f32)  # (T, 3, 3)
    if has_n:
        N = np.array([[nrm[v[2]] for v in t] for t in tris], f32)
    else:
        acc = {}
        for t, tri in enumerate(tris):
            v0, v1, v2 = P[t]
            e1, e2 = v1 - v0, v2 - v1
            n = np.array([e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0],
                          e1[0] * e2[1] - e2[0] * e1[1]], f32)
            for v in tri:
                acc[v[0]] = acc.get(v[0], np.zeros(3, f32)) + n
        for k, n in acc.items():
            ln = np.sqrt(f32(f32(n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]))
            if ln > 0:
                acc[k] = n / ln
        N = np.array([[acc[v[0]] for v in t] for t in tris], f32)
    if has_t:
        UV = np.array([[tex[v[1]] for v in t] for t in tris], f32)
    else:
        UV = np.zeros((len(tris), 3, 2), f32)
    return P, N, UV


# --------------------------------------------------------------------------
# bvh.rs
# --------------------------------------------------------------------------
def _half_area(mn, mx):
    with np.errstate(all="ignore"):
        e = mx - mn
        return f32(f32(e[0] * e[1] + e[1] * e[2]) + e[0] * e[2])


def fmin_self(a, b):
    """f32::min as rustc compiles it on x86-64 (llvm.minnum -> `b < a ? b : a`, NaN self replaced by b): among
    equal operands -- zeros of either sign -- SELF wins.  The Rust docs leave the zero sign open; the product's
    C++ builder is compiled by the same LLVM lowering (std::fmin under clang), numpy's minimum would let the
    second operand win."""
    with np.errstate(all="ignore"):
        return np.where((b < a) | np.isnan(a), b, a).astype(np.float32)


def fmax_self(a, b):
    with np.errstate(all="ignore"):
        return np.where((b > a) | np.isnan(a), b, a).astype(np.float32)


def _seq_min(v, init):
    """fold of fmin_self over the rows of v in order, starting from `init` (bvh.rs:292-297 fit_bounds)."""
    if len(v) == 0:
        return np.full(3, init, f32)
    m = v.min(0)
    out = fmin_self(np.full(3, init, f32), m)
    for k in range(3):
        if out[k] == 0:   # the first zero met keeps its sign
            z = v[v[:, k] == 0, k]
            if len(z):
                out[k] = z[0]
    return out


def _seq_max(v, init):
    if len(v) == 0:
        return np.full(3, init, f32)
    m = v.max(0)
    out = fmax_self(np.full(3, init, f32), m)
    for k in range(3):
        if out[k] == 0:
            z = v[v[:, k] == 0, k]
            if len(z):
                out[k] = z[0]
    return out


def build_bvh(P):
    """P: (T, 3, 3) float32 triangle positions.  Returns (order, nodes) with
    nodes as dict rows (left, right, first, count, aabb_min, aabb_max) and
    `order` the permutation of triangles (bvh.rs:208-290, Quality::High)."""
    T = P.shape[0]
    cen = ((P[:, 0] + P[:, 1]) + P[:, 2]) * f32(1.0 / 3.0)
    tmin = fmin_self(P[:, 0], fmin_self(P[:, 1], P[:, 2]))   # bvh.rs:237-238: v1.min(v2.min(v3))
    tmax = fmax_self(P[:, 0], fmax_self(P[:, 1], P[:, 2]))
    order = np.arange(T)
    big0, small0 = f32(np.finfo(np.float32).max), f32(np.finfo(np.float32).min)
    nodes = [dict(left=0, right=0, first=0, count=T, mn=_seq_min(tmin, big0), mx=_seq_max(tmax, small0))]

    def evaluate_sah(axis, pos, start, count):
        idx = order[start:start + count]
        left = cen[idx, axis] < pos
        nl, nr = int(left.sum()), int(count - left.sum())
        inf = f32(np.inf)
        lmn = tmin[idx[left]].min(0) if nl else np.full(3, inf, f32)
        lmx = tmax[idx[left]].max(0) if nl else np.full(3, -inf, f32)
        rmn = tmin[idx[~left]].min(0) if nr else np.full(3, inf, f32)
        rmx = tmax[idx[~left]].max(0) if nr else np.full(3, -inf, f32)
        with np.errstate(all="ignore"):
            return f32(f32(nl) * _half_area(lmn, lmx) + f32(nr) * _half_area(rmn, rmx))

    def subdivide(ni, start, n, depth):
        nd = nodes[ni]
        e = nd["mx"] - nd["mn"]
        parent_cost = f32(f32(f32(e[0] * e[1] + e[1] * e[2]) + e[0] * e[2]) * f32(nd["count"]))
        best, axis, split = f32(np.inf), 0, f32(0)
        if nd["count"] > 1:
            max_axis = max(e[0], max(e[1], e[2]))
            for a in range(3):
                if e[a] == 0:
                    continue
                with np.errstate(all="ignore"):
                    c = math.ceil(f32(f32(e[a] / max_axis) * f32(50.0)))
                nt = min(max(int(c), 1), 50)
                for i in range(nt):
                    st = f32(f32(i + 1) / f32(f32(nt) + f32(1.0)))
                    pos = f32(nd["mn"][a] + f32(e[a] * st))
                    cost = evaluate_sah(a, pos, start, n)
                    if cost < best:
                        best, axis, split = cost, a, pos
        if best < parent_cost and depth < 32:
            lc = 0
            big, small = f32(np.finfo(np.float32).max), f32(np.finfo(np.float32).min)
            lmn, lmx = np.full(3, big, f32), np.full(3, small, f32)
            rmn, rmx = np.full(3, big, f32), np.full(3, small, f32)
            for i in range(start, start + n):
                t = order[i]
                if cen[t, axis] < split:
                    lmn, lmx = fmin_self(lmn, tmin[t]), fmax_self(lmx, tmax[t])
                    order[start + lc], order[i] = order[i], order[start + lc]
                    lc += 1
                else:
                    rmn, rmx = fmin_self(rmn, tmin[t]), fmax_self(rmx, tmax[t])
            li, ri = len(nodes), len(nodes) + 1
            nodes.append(dict(left=0, right=0, first=start, count=lc, mn=lmn, mx=lmx))
            nodes.append(dict(left=0, right=0, first=start + lc, count=n - lc, mn=rmn, mx=rmx))
            nd["left"], nd["right"], nd["count"] = li, ri, 0
            subdivide(li, start, lc, depth + 1)
            subdivide(ri, start + lc, n - lc, depth + 1)

    subdivide(0, 0, T, 0)
    return order, nodes
