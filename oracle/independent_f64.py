"""independent_f64.py -- TEST INFRASTRUCTURE: a second, independent restatement of the path-trace loop of
/root/reference/shaders/ray_tracer.wgsl in float64 numpy, for mesh scenes without glass, textures or spheres (the
Cornell box of BASELINE configs[0..1]).

Why it exists.  oracle/shader_oracle.cpp and the HIP kernels compile the same headers (csrc/rt_transc.h,
rt_texture.h): the polynomials for log / sin / cos / pow, the two-step normalize and the rounding of every operation
are DEFINED there, so the kernels' bit-for-bit parity with the oracle cannot notice a wrong definition.  This file
shares nothing with them: double precision, numpy / libm `log`, `cos`, `sin`, `sqrt`, `power`, true division, the
shader's literal constants, and no BVH at all (every triangle of every mesh is tested: in exact arithmetic the BVH
only skips triangles that cannot be the closest hit).  It pins the canonical float32 arithmetic STATISTICALLY -- the
closest available stand-in for north_star's "within 1e-5 of the reference render", which cannot be evaluated here
(the reference cannot be built or run: SURVEY.md section 8c) -- and guards later redefinitions of that arithmetic.

Only tests/ may import this module (and tests/golden/make_f64_pin.py, which writes the committed statistics).
PARITY UNPINNED like the rest of oracle/ (no vectors exist in the reference).

Each function cites the wgsl lines it follows (all `file:line` relative to /root/reference/shaders/ray_tracer.wgsl).
"""
import numpy as np

EPSILON = 1e-5                     # :131
INF = float.fromhex("0x1p+127")    # :132
PI = 3.1415926                     # the shader's literal (:182,203)
SKY_HORIZON = np.array([1.0, 1.0, 1.0, 0.0])                   # :126
SKY_ZENITH = np.array([0.0788092, 0.36480793, 0.7264151, 0.0])  # :127
GROUND_COLOR = np.array([0.35, 0.3, 0.35, 0.0])                # :128


class Rng:
    """:195-200 on a vector of u32 states (exact integer arithmetic); rand :164-166 as a double division."""

    def __init__(self, state):
        self.s = state.astype(np.uint64)

    def next(self, mask=None):
        m = np.uint64(0xffffffff)
        s = (self.s * np.uint64(747796405) + np.uint64(2891336453)) & m
        if mask is not None:
            s = np.where(mask, s, self.s)
        self.s = s
        r = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & m
        return (r >> np.uint64(22)) ^ r

    def rand(self, mask=None):
        return self.next(mask).astype(np.float64) / 4294967295.0   # :165 (2^32 - 1, exactly, in double)

    def normal(self, mask=None):   # :181-185
        theta = 2.0 * PI * self.rand(mask)
        with np.errstate(divide="ignore"):
            rho = np.sqrt(-2.0 * np.log(self.rand(mask)))
        return rho * np.cos(theta)

    def unit_sphere(self, mask=None):   # :168-174
        x = self.normal(mask)
        y = self.normal(mask)
        z = self.normal(mask)
        return normalize(np.stack([x, y, z], -1))

    def in_unit_disk(self, mask=None):   # :202-206
        angle = self.rand(mask) * 2.0 * PI
        r = np.sqrt(self.rand(mask))
        return np.cos(angle) * r, np.sin(angle) * r


def dot(a, b):
    return (a * b).sum(-1)


def normalize(v):
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.sqrt(dot(v, v))[..., None]


def smoothstep(lo, hi, x):   # WGSL builtin
    t = np.clip((x - lo) / (hi - lo), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)


def mix(a, b, t):
    return a * (1.0 - t) + b * t


def environment_light(d):   # :214-221
    y = d[:, 1]
    sky_t = np.power(smoothstep(0.0, 0.4, y), 0.35)
    g2s = smoothstep(-0.01, 0.0, y)
    sky = mix(SKY_HORIZON[None, :], SKY_ZENITH[None, :], sky_t[:, None])
    sun = np.power(np.maximum(0.0, dot(d, np.array([0.1, 1.0, 0.1]))), 500.0) * 0.1
    return mix(GROUND_COLOR[None, :], sky, g2s[:, None]) + (sun * (g2s >= 1.0))[:, None]


class Scene:
    """The reference's arrays (MeshUniform / PackedTriangle, include/rt_abi.h section 1) as doubles.  Triangles are
    taken per mesh by walking the mesh's BVH nodes for their leaf ranges only -- the boxes are never used."""

    def __init__(self, arrays):
        assert arrays.spheres.shape[0] == 0, "meshes only"
        self.cam_to_world = np.array(arrays.uniform.camera.cam_to_world, np.float64)   # [col][row]
        self.view_params = np.array(arrays.uniform.camera.view_params, np.float64)
        self.defocus = float(arrays.uniform.camera.defocus_strength)
        self.diverge = float(arrays.uniform.camera.diverge_strength)
        self.meshes = []
        t = arrays.triangles
        for m in arrays.meshes:
            mat = m["material"]
            assert int(mat["flag"]) == 0, "no glass, no textures"
            idx = self._leaf_triangles(arrays.nodes, int(m["node_offset"]), int(m["triangle_offset"]))
            f = lambda k: t[k][idx].astype(np.float64)  # noqa: E731
            self.meshes.append(dict(
                w2m=np.array(m["world_to_model"], np.float64), m2w=np.array(m["model_to_world"], np.float64),
                v1=f("v1"), v2=f("v2"), v3=f("v3"), n1=f("n1"), n2=f("n2"), n3=f("n3"),
                uv1=np.stack([f("uv10"), f("uv11")], -1), uv2=np.stack([f("uv20"), f("uv21")], -1),
                uv3=np.stack([f("uv30"), f("uv31")], -1),
                color=np.array(mat["color"], np.float64), emission=np.array(mat["emission_color"], np.float64),
                specular_color=np.array(mat["specular_color"], np.float64), emission_strength=float(mat["emission_strength"]),
                smoothness=float(mat["smoothness"]), specular=float(mat["specular"])))

        for k in ("specular", "smoothness", "emission_strength", "emission", "color", "specular_color"):
            setattr(self, "mat_" + k, np.array([m[k] for m in self.meshes], np.float64))

    @staticmethod
    def _leaf_triangles(nodes, node_offset, tri_offset):
        out, st = [], [0]
        while st:
            n = nodes[node_offset + st.pop()]
            if n["count"] > 0:
                out.extend(range(tri_offset + int(n["first"]), tri_offset + int(n["first"]) + int(n["count"])))
            else:
                st.extend([int(n["right"]), int(n["left"])])
        return np.array(out, np.int64)


def mat_point(m, v, w):   # (mat4 * vec4(v, w)).xyz, m[col][row]
    return v[:, 0:1] * m[0, :3] + v[:, 1:2] * m[1, :3] + v[:, 2:3] * m[2, :3] + w * m[3, :3]


def closest_hit(scene, ro, rd):
    """calculate_ray_collions :353-396 with ray_triangle :258-290 over ALL triangles of a mesh.  Returns hit mask, world
    distance, world hit point, world normal, uv, mesh index."""
    n = ro.shape[0]
    best = np.full(n, INF)
    hit = np.zeros(n, bool)
    point = np.zeros((n, 3))
    normal = np.zeros((n, 3))
    uv = np.zeros((n, 2))
    which = np.full(n, -1)
    for mi, m in enumerate(scene.meshes):
        lo = mat_point(m["w2m"], ro, 1.0)                  # :371
        ld = normalize(mat_point(m["w2m"], rd, 0.0))       # :372
        eab, eac = m["v2"] - m["v1"], m["v3"] - m["v1"]    # :261-262
        nrm = np.cross(eab, eac)                           # :263
        ao = lo[:, None, :] - m["v1"][None, :, :]          # :264
        dao = np.cross(ao, ld[:, None, :])                 # :265
        det = -(ld[:, None, :] * nrm[None]).sum(-1)        # :266
        keep = det >= 1e-8                                 # :268 (cull_backface: no glass here)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv = 1.0 / det
            dst = (ao * nrm[None]).sum(-1) * inv
            u = (eac[None] * dao).sum(-1) * inv
            v = -(eab[None] * dao).sum(-1) * inv
        w = 1.0 - u - v
        ok = keep & (dst > EPSILON) & (u >= 0.0) & (v >= 0.0) & (w >= 0.0)   # :280
        t = np.where(ok, dst, np.inf)
        k = t.argmin(1)                                    # closest triangle of the mesh (ray_BVH keeps strictly closer hits)
        r = np.arange(n)
        mesh_hit = np.isfinite(t[r, k])
        tk, uk, vk, wk = t[r, k], u[r, k], v[r, k], w[r, k]
        ln = normalize(m["n1"][k] * wk[:, None] + m["n2"][k] * uk[:, None] + m["n3"][k] * vk[:, None])   # :282 (det > 0: sign = 1)
        luv = m["uv1"][k] * wk[:, None] + m["uv2"][k] * uk[:, None] + m["uv3"][k] * vk[:, None]
        lhp = lo + ld * np.where(mesh_hit, tk, 0.0)[:, None]      # :379
        whp = mat_point(m["m2w"], lhp, 1.0)                       # :380
        d = ro - whp
        wdst = np.sqrt(dot(d, d))                                 # :381
        better = mesh_hit & (wdst < best)                         # :383
        best = np.where(better, wdst, best)
        hit |= better
        point = np.where(better[:, None], whp, point)
        normal = np.where(better[:, None], normalize(mat_point(m["m2w"], ln, 0.0)), normal)   # :386
        uv = np.where(better[:, None], luv, uv)
        which = np.where(better, mi, which)
    return hit, best, point, normal, uv, which


def primary_rays(scene, W, H, rng=None):
    """frag :473-495 for every pixel (x fastest); with rng: the four jitter draws per sample."""
    y, x = np.mgrid[0:H, 0:W]
    px = np.stack([x.reshape(-1), y.reshape(-1)], -1).astype(np.float64)
    uv = px / (np.array([W, H], np.float64) - 1.0)                          # :479
    c2w = scene.cam_to_world
    origin = np.broadcast_to(c2w[3, :3], (W * H, 3)).copy()
    local = np.concatenate([uv - 0.5, np.ones((W * H, 1))], -1) * scene.view_params   # :481
    focus = mat_point(c2w, local, 1.0)                                       # :482
    if rng is not None:
        right, up = c2w[0, :3], c2w[1, :3]
        jx, jy = rng.in_unit_disk()                                          # :488
        origin = origin + right * (jx * scene.defocus / W)[:, None] + up * (jy * scene.defocus / W)[:, None]
        kx, ky = rng.in_unit_disk()                                          # :492
        focus = focus + right * (kx * scene.diverge / W)[:, None] + up * (ky * scene.diverge / W)[:, None]
    return origin, normalize(focus - origin)                                 # :494


def render_frame(scene, W, H, bounces, spp, frames, skybox=1):
    """`frag` :473-500 + `trace` :398-471 for one frame: the per-frame sample image (H, W, 4) in float64."""
    y, x = np.mgrid[0:H, 0:W]
    seed = (y.reshape(-1).astype(np.float64) * W + x.reshape(-1)).astype(np.uint64) + np.uint64(abs(frames)) * np.uint64(719393)   # :475
    rng = Rng(seed & np.uint64(0xffffffff))
    n = W * H
    total = np.zeros((n, 4))
    for _ in range(spp):
        ro, rd = primary_rays(scene, W, H, rng)
        rd = normalize(rd)                                  # :400
        T = np.ones((n, 4))
        light = np.zeros((n, 4))
        alive = np.ones(n, bool)
        for _seg in range(bounces + 1):                     # :404
            if not alive.any():
                break
            idx = np.nonzero(alive)[0]
            hit, dst, point, normal, _uv, which = closest_hit(scene, ro[idx], rd[idx])
            miss = idx[~hit]
            if skybox and miss.size:                        # :406-411
                light[miss] += T[miss] * environment_light(rd[miss])
            alive[miss] = False
            h = idx[hit]
            if h.size == 0:
                break
            hm = np.zeros(n, bool)
            hm[h] = True
            wm = which[hit]
            spec, smooth, es = scene.mat_specular[wm], scene.mat_smoothness[wm], scene.mat_emission_strength[wm]
            ecol, col, scol = scene.mat_emission[wm], scene.mat_color[wm], scene.mat_specular_color[wm]
            nrm = normal[hit]
            ro[h] = point[hit]                              # :413
            is_spec = spec >= rng.rand(hm)[h]               # :438
            sph = rng.unit_sphere(hm)[h]                    # :448 (rand_hemisphere :176-179)
            diffuse = sph * np.sign(dot(nrm, sph))[:, None]
            specular_dir = rd[h] - 2.0 * dot(nrm, rd[h])[:, None] * nrm      # reflect :449
            light[h] += ecol * es[:, None] * T[h]           # :452 (before the albedo multiply)
            rd[h] = normalize(mix(diffuse, specular_dir, (smooth * is_spec)[:, None]))   # :451
            T[h] *= np.where(is_spec[:, None], scol, col)   # :459
            p = T[h, :3].max(-1)                            # :462
            die = rng.rand(hm)[h] >= p                      # :463
            alive[h[die]] = False
            live = h[~die]
            with np.errstate(divide="ignore", invalid="ignore"):
                T[live] *= (1.0 / p[~die])[:, None]         # :466
        total += light                                      # :496
    return (total / spp).reshape(H, W, 4)                   # :498


def debug_view(scene, W, H, mode, scale):
    """debug_trace :502-573 for the views that do not count BVH tests: 1 normals, 2 depth, 3 texcoords, 4 focus distance."""
    ro, rd = primary_rays(scene, W, H)
    hit, dst, _point, normal, uv, _which = closest_hit(scene, ro, rd)
    out = np.zeros((W * H, 4))
    if mode == 1:
        out[hit] = np.concatenate([normal[hit] * 0.5 + 0.5, np.ones((hit.sum(), 1))], -1)
    elif mode == 2:
        d = dst[hit] / float(scale)
        out[hit] = np.stack([d, d, d, np.ones_like(d)], -1)
    elif mode == 3:
        out[hit] = np.concatenate([uv[hit], np.zeros((hit.sum(), 1)), np.ones((hit.sum(), 1))], -1)
    elif mode == 4:
        s, d = float(scale) / 100.0, dst[hit]
        out[hit] = np.where((d > s)[:, None], np.array([0.0, 1.0, 0.0, 1.0]), np.stack([d, d, d, np.ones_like(d)], -1))
    else:
        raise ValueError("views 5-7 count BVH tests: this restatement has no BVH")
    return out.reshape(H, W, 4), hit.reshape(H, W)
