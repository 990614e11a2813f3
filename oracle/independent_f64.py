"""independent_f64.py -- TEST INFRASTRUCTURE: a second, independent restatement of the path-trace loop of
/root/reference/shaders/ray_tracer.wgsl in float64 numpy: meshes, spheres, glass, textured materials, depth of field (the
Cornell box of BASELINE configs[0..1] and the scene library's `room`, `metal`, `balls`, `texture_test`).

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


MAT_FIELDS = ("color", "emission_color", "specular_color", "absorption", "absorption_strength", "emission_strength", "smoothness",
              "specular", "ior", "flag", "diffuse_index")


class Scene:
    """The reference's arrays (MeshUniform / PackedTriangle / Sphere / textures, include/rt_abi.h section 1) as doubles.
    Triangles are taken per mesh by walking the mesh's BVH nodes for their leaf ranges only -- the boxes are never used.
    Materials: mesh materials first, then the spheres' (index = number of meshes + sphere index)."""

    def __init__(self, arrays):
        self.cam_to_world = np.array(arrays.uniform.camera.cam_to_world, np.float64)   # [col][row]
        self.view_params = np.array(arrays.uniform.camera.view_params, np.float64)
        self.defocus = float(arrays.uniform.camera.defocus_strength)
        self.diverge = float(arrays.uniform.camera.diverge_strength)
        self.meshes = []
        t = arrays.triangles
        mats = []
        for m in arrays.meshes:
            idx = self._leaf_triangles(arrays.nodes, int(m["node_offset"]), int(m["triangle_offset"]))
            f = lambda k: t[k][idx].astype(np.float64)  # noqa: E731
            self.meshes.append(dict(
                w2m=np.array(m["world_to_model"], np.float64), m2w=np.array(m["model_to_world"], np.float64),
                v1=f("v1"), v2=f("v2"), v3=f("v3"), n1=f("n1"), n2=f("n2"), n3=f("n3"),
                uv1=np.stack([f("uv10"), f("uv11")], -1), uv2=np.stack([f("uv20"), f("uv21")], -1),
                uv3=np.stack([f("uv30"), f("uv31")], -1), glass=int(m["material"]["flag"]) == 1))
            mats.append(m["material"])
        self.sphere_pos = np.array([sp["pos"] for sp in arrays.spheres], np.float64).reshape(-1, 3)
        self.sphere_radius = np.array([sp["radius"] for sp in arrays.spheres], np.float64)
        mats += [sp["material"] for sp in arrays.spheres]
        self.n_meshes = len(self.meshes)
        for k in MAT_FIELDS:
            setattr(self, "mat_" + k, np.array([np.asarray(mm[k]) for mm in mats], np.float64 if k not in ("flag", "diffuse_index") else np.int64))
        self.textures = [np.asarray(tex, np.uint8) for tex in arrays.textures]   # (H, W, 4) sRGB, already flipped by the loader

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


def srgb_to_linear(c8):
    """The sRGB decode of an Rgba8UnormSrgb texture fetch (src/rendering/ray_tracer.rs:253), from the IEC 61966-2-1 formula."""
    c = c8.astype(np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, np.power((c + 0.055) / 1.055, 2.4))


def sample_texture(tex, u, v):
    """textureSampleLevel(.., uv, 0.0) with the reference's sampler (src/rendering/ray_tracer.rs:197-205): bilinear, repeat,
    texel centres at (i + 0.5) / size, sRGB decoded before filtering, alpha linear.  wgsl:455."""
    h, w = tex.shape[:2]
    px, py = u * w - 0.5, v * h - 0.5
    x0, y0 = np.floor(px), np.floor(py)
    fx, fy = (px - x0)[:, None], (py - y0)[:, None]
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def texel(x, y):
        t = tex[np.mod(y, h), np.mod(x, w)]
        return np.concatenate([srgb_to_linear(t[:, :3]), t[:, 3:4].astype(np.float64) / 255.0], -1)
    top = texel(x0, y0) * (1.0 - fx) + texel(x0 + 1, y0) * fx
    bot = texel(x0, y0 + 1) * (1.0 - fx) + texel(x0 + 1, y0 + 1) * fx
    return top * (1.0 - fy) + bot * fy


def mat_point(m, v, w):   # (mat4 * vec4(v, w)).xyz, m[col][row]
    return v[:, 0:1] * m[0, :3] + v[:, 1:2] * m[1, :3] + v[:, 2:3] * m[2, :3] + w * m[3, :3]


def closest_hit(scene, ro, rd):
    """calculate_ray_collions :353-396: ray_sphere :223-256 over the spheres, then ray_triangle :258-290 over ALL triangles
    of every mesh.  Returns hit mask, world distance, world hit point, world normal, uv, material index, backface."""
    n = ro.shape[0]
    best = np.full(n, INF)
    hit = np.zeros(n, bool)
    point = np.zeros((n, 3))
    normal = np.zeros((n, 3))
    uv = np.zeros((n, 2))
    which = np.full(n, -1)
    backface = np.zeros(n, bool)
    for si in range(scene.sphere_pos.shape[0]):             # :359-367
        oc = ro - scene.sphere_pos[si]
        a = dot(rd, rd)
        b = 2.0 * dot(oc, rd)
        c = dot(oc, oc) - scene.sphere_radius[si] ** 2
        disc = b * b - 4.0 * a * c
        ok = disc >= 0.0
        with np.errstate(invalid="ignore", divide="ignore"):
            sq = np.sqrt(np.where(ok, disc, 0.0))
            near = np.maximum(0.0, (-b - sq) / (2.0 * a))
            far = (-b + sq) / (2.0 * a)
        ok &= far >= 0.001
        inside = near == 0.0
        dst = np.where(inside, far, near)
        better = ok & (dst < best)                          # :362
        hp = ro + rd * dst[:, None]
        nrm = normalize(hp - scene.sphere_pos[si])
        nrm = np.where(inside[:, None], -nrm, nrm)
        with np.errstate(invalid="ignore"):
            theta = np.arccos(np.clip(-nrm[:, 1], -1.0, 1.0))
        phi = np.arctan2(-nrm[:, 2], -nrm[:, 0]) + PI
        suv = np.stack([phi / (2.0 * PI), theta / PI], -1)
        best = np.where(better, dst, best)
        hit |= better
        point = np.where(better[:, None], hp, point)
        normal = np.where(better[:, None], nrm, normal)
        uv = np.where(better[:, None], suv, uv)
        which = np.where(better, scene.n_meshes + si, which)
        backface = np.where(better, inside, backface)
    for mi, m in enumerate(scene.meshes):
        lo = mat_point(m["w2m"], ro, 1.0)                  # :371
        ld = normalize(mat_point(m["w2m"], rd, 0.0))       # :372
        eab, eac = m["v2"] - m["v1"], m["v3"] - m["v1"]    # :261-262
        nrm = np.cross(eab, eac)                           # :263
        ao = lo[:, None, :] - m["v1"][None, :, :]          # :264
        dao = np.cross(ao, ld[:, None, :])                 # :265
        det = -(ld[:, None, :] * nrm[None]).sum(-1)        # :266
        keep = (np.abs(det) >= 1e-8) if m["glass"] else (det >= 1e-8)   # :268 (cull_backface = not glass, :375)
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
        tk, uk, vk, wk, dk = t[r, k], u[r, k], v[r, k], w[r, k], det[r, k]
        ln = normalize(m["n1"][k] * wk[:, None] + m["n2"][k] * uk[:, None] + m["n3"][k] * vk[:, None]) * np.sign(dk)[:, None]   # :282
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
        backface = np.where(better, dk < 0.0, backface)           # :283
    return hit, best, point, normal, uv, which, backface


def refract(I, N, eta):   # WGSL builtin
    d = dot(N, I)
    k = 1.0 - eta * eta * (1.0 - d * d)
    with np.errstate(invalid="ignore"):
        r = eta[:, None] * I - (eta * d + np.sqrt(np.maximum(k, 0.0)))[:, None] * N
    return np.where((k < 0.0)[:, None], 0.0, r)


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
            hit, dst, point, normal, uvh, which, backface = closest_hit(scene, ro[idx], rd[idx])
            miss = idx[~hit]
            if skybox and miss.size:                        # :406-411
                light[miss] += T[miss] * environment_light(rd[miss])
            alive[miss] = False
            h = idx[hit]
            if h.size == 0:
                break
            wm, nrm, dsth, bf, uvm = which[hit], normal[hit], dst[hit], backface[hit], uvh[hit]
            ro[h] = point[hit]                              # :413
            glass = scene.mat_flag[wm] == 1
            # ---- glass :414-436 ----
            g = np.nonzero(glass)[0]
            if g.size:
                hg = h[g]
                gm = np.zeros(n, bool)
                gm[hg] = True
                mi = wm[g]
                Tg = T[hg].copy()
                absorb = np.exp(-dsth[g][:, None] * scene.mat_absorption[mi][:, :3] * scene.mat_absorption_strength[mi][:, None])
                Tg_in = np.concatenate([Tg[:, :3] * absorb, np.ones((g.size, 1))], -1)
                Tg = np.where(bf[g][:, None], Tg_in, Tg)                                    # :415-418
                ior = np.where(bf[g], scene.mat_ior[mi], 1.0 / scene.mat_ior[mi])           # :420
                d_in, ng = rd[hg], nrm[g]
                refl = d_in - 2.0 * dot(ng, d_in)[:, None] * ng                             # :422
                refr = refract(d_in, ng, ior)                                               # :423
                cos_t = np.minimum(dot(-d_in, ng), 1.0)
                sin_t = np.sqrt(1.0 - cos_t * cos_t)
                cannot = ior * sin_t > 1.0
                r0 = ((1.0 - ior) / (1.0 + ior)) ** 2
                schlick = r0 + (1.0 - r0) * np.power(1.0 - cos_t, 5.0)                       # :208-212
                draw = np.zeros(n, bool)
                draw[hg[~cannot]] = True                                                    # `||` short-circuits: no draw when cannot_refract (:428)
                follow = cannot | (schlick > rng.rand(draw)[hg])
                diffuse = normalize(ng + rng.unit_sphere(gm)[hg])                           # :430 (rand_direction :187-193)
                refl = normalize(mix(diffuse, refl, scene.mat_specular[mi][:, None]))       # :432
                refr = normalize(mix(-diffuse, refr, scene.mat_smoothness[mi][:, None]))    # :433
                nd = np.where(follow[:, None], refl, refr)
                rd[hg] = nd
                ro[hg] = point[hit][g] + 1e-4 * ng * np.sign(dot(ng, nd))[:, None]          # :436
                T[hg] = Tg
            # ---- everything else :437-460 ----
            o = np.nonzero(~glass)[0]
            if o.size:
                ho = h[o]
                om = np.zeros(n, bool)
                om[ho] = True
                mi = wm[o]
                is_spec = scene.mat_specular[mi] >= rng.rand(om)[ho]            # :438
                sph = rng.unit_sphere(om)[ho]                                   # :448 (rand_hemisphere :176-179)
                no = nrm[o]
                diffuse = sph * np.sign(dot(no, sph))[:, None]
                specular_dir = rd[ho] - 2.0 * dot(no, rd[ho])[:, None] * no     # reflect :449
                light[ho] += scene.mat_emission_color[mi] * scene.mat_emission_strength[mi][:, None] * T[ho]   # :452
                rd[ho] = normalize(mix(diffuse, specular_dir, (scene.mat_smoothness[mi] * is_spec)[:, None]))   # :451
                col = scene.mat_color[mi].copy()
                tex = (scene.mat_flag[mi] == 2) & (scene.mat_diffuse_index[mi] != -1)                         # :454
                for ti in np.unique(scene.mat_diffuse_index[mi][tex]):
                    sel = tex & (scene.mat_diffuse_index[mi] == ti)
                    col[sel] = sample_texture(scene.textures[int(ti)], uvm[o][sel, 0], uvm[o][sel, 1])         # :455
                T[ho] *= np.where(is_spec[:, None], scene.mat_specular_color[mi], col)                         # :459
            hm = np.zeros(n, bool)
            hm[h] = True
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
    hit, dst, _point, normal, uv, _which, _bf = closest_hit(scene, ro, rd)
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
