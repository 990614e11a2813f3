"""Host-side scene pipeline (C++: OBJ/MTL loader, SAH BVH, scene library)
against the independent Python restatement oracle/host_oracle.py, against the
committed Cornell fixture, and against the facts pinned in SURVEY.md 8a."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE_ASSETS, ROOT, bits

DATA = os.path.join(ROOT, "tests", "data")


def test_tobj_model_splitting_rules(rt):
    sc = rt.Scene()
    sc.add_mesh_from_file("quirks.obj", use_mtl=True, assets_dir=DATA)
    sc.build()
    # `usemtl` with a new material flushes a model under the *current* name; `g`/`o`
    # after the faces names the *next* model; pentagon -> 3-triangle fan
    assert sc.mesh_labels() == ["first", "first", "late_name", "third"]
    m = sc.meshes()
    assert m["triangles"].tolist() == [2, 2, 3, 1]
    red, glassy = m["material"][0], m["material"][1]
    assert np.allclose(red["color"], [0.9, 0.1, 0.2, 1.0]) and red["flag"] == 0
    assert red["specular"] == np.float32(0.5) and red["smoothness"] == np.float32(np.sqrt(np.float32(0.64)))
    assert red["emission_strength"] == np.float32(6.0)          # 2 * max(Ke)
    assert np.allclose(red["emission_color"], [1.0, 0.5, 0.25, 1.0])
    assert glassy["flag"] == 1 and glassy["ior"] == np.float32(1.5)   # illum 4 -> GLASS
    assert glassy["specular"] == np.float32(1.0) and glassy["smoothness"] == 0   # Ks defaults to 1, Ns to 0
    assert m["material"][2]["flag"] == 1 and m["material"][3]["flag"] == 0
    t = sc.triangles()
    # model 0 has vt + vn; its first triangle keeps the file's uv and normal
    k = int(np.argmin(np.abs(t["v1"][:2] - np.float32([0, 0, 0])).sum(1)))
    assert t["n1"][k].tolist() == [0, 0, 1]
    # model 2 (no vn): synthesised normals are unit length
    tri2 = t[m["triangle_offset"][2]:m["triangle_offset"][2] + 3]
    assert np.allclose(np.linalg.norm(tri2["n1"], axis=1), 1.0, atol=1e-6)


def test_negative_indices_are_relative(rt):
    sc = rt.Scene()
    sc.add_mesh_from_file("quirks.obj", use_mtl=False, assets_dir=DATA, mat=rt.material())
    sc.build()
    t = sc.triangles()
    off = sc.meshes()["triangle_offset"][1]
    zs = np.concatenate([t["v1"][off:off + 2, 2], t["v2"][off:off + 2, 2], t["v3"][off:off + 2, 2]])
    assert np.all(zs == 1.0)   # `f -3//1 -2//1 -1//1` addressed the three z = 1 vertices


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ASSETS), reason="reference assets not present")
def test_cornell_from_assets_matches_fixture_and_survey(rt):
    sc = rt.Scene.from_name("cornell_box", REFERENCE_ASSETS)
    arr = rt.SceneArrays.from_scene(sc)
    fix = rt.SceneArrays.load(os.path.join(GOLDEN, "cornell_scene.npz"))
    assert arr.meshes.tobytes() == fix.meshes.tobytes()
    assert arr.triangles.tobytes() == fix.triangles.tobytes()
    assert arr.nodes.tobytes() == fix.nodes.tobytes()
    assert bytes(arr.uniform) == bytes(fix.uniform)
    # 8 tobj models, two of them named leftWall; nodes per mesh 1,1,3,1,3,11,11,1
    assert sc.mesh_labels() == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "leftWall", "shortBox", "light"]
    assert arr.meshes["triangles"].tolist() == [2, 2, 2, 2, 2, 10, 10, 2]
    offs = arr.meshes["node_offset"].tolist() + [32]
    assert [b - a for a, b in zip(offs, offs[1:])] == [1, 1, 3, 1, 3, 11, 11, 1]


def test_cornell_fixture_facts(cornell):
    u = cornell.uniform
    assert (u.spheres, u.meshes, u.nodes, u.n_indices) == (0, 8, 32, 96)
    # view_params = (3.5555556, 2, 1); cam_to_world rotation diag(-1, 1, -1), translation (0, 1, 2)
    assert list(u.camera.view_params) == [np.float32(3.5555556), 2.0, 1.0]
    c = np.array([list(col) for col in u.camera.cam_to_world])
    assert np.array_equal(c[:3, :3], np.diag([-1.0, 1.0, -1.0])) and c[3].tolist() == [0, 1, 2, 1]
    # every wall specular 0, smoothness sqrt(0.1); light emission (1, 12/17, 4/17) x 34, albedo 0.78
    m = cornell.meshes["material"]
    assert np.all(m["specular"] == 0) and np.all(m["smoothness"] == np.float32(np.sqrt(np.float32(0.1))))
    assert np.all(m["flag"] == 0)
    assert m["emission_strength"][7] == 34 and np.allclose(m["emission_color"][7], [1, 12 / 17, 4 / 17, 1], atol=1e-7)
    assert np.allclose(m["color"][7][:3], 0.78)
    assert np.all(cornell.meshes["model_to_world"] == np.eye(4, dtype=np.float32))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ASSETS), reason="reference assets not present")
def test_loader_against_python_oracle(rt):
    from oracle import host_oracle as ho
    models, mats, pos, tex, nrm = ho.load_obj(os.path.join(REFERENCE_ASSETS, "CornellBox-Original.obj"))
    assert [m["name"] for m in models] == ["floor", "ceiling", "backWall", "rightWall", "leftWall", "leftWall", "shortBox", "light"]
    fix = rt.SceneArrays.load(os.path.join(GOLDEN, "cornell_scene.npz"))
    for i, model in enumerate(models):
        mat = ho.material_from_mtl(mats[model["material_id"]])
        got = fix.meshes["material"][i]
        for k, v in mat.items():
            assert np.array_equal(np.float32(got[k]) if k != "flag" else got[k], np.float32(v) if k != "flag" else v), (i, k)
        P, N, UV = ho.unroll_model(model, pos, tex, nrm)
        order, nodes = ho.build_bvh(P)
        t0 = fix.meshes["triangle_offset"][i]
        tri = fix.triangles[t0:t0 + len(order)]
        assert np.array_equal(bits(tri["v1"]), bits(P[order, 0])) and np.array_equal(bits(tri["v3"]), bits(P[order, 2]))
        assert np.array_equal(bits(tri["n2"]), bits(N[order, 1]))
        n0 = fix.meshes["node_offset"][i]
        nd = fix.nodes[n0:n0 + len(nodes)]
        assert nd["left"].tolist() == [n["left"] for n in nodes] and nd["count"].tolist() == [n["count"] for n in nodes]
        assert nd["first"].tolist() == [n["first"] for n in nodes]
        assert np.array_equal(bits(nd["aabb_min"]), bits(np.array([n["mn"] for n in nodes], np.float32)))
        assert np.array_equal(bits(nd["aabb_max"]), bits(np.array([n["mx"] for n in nodes], np.float32)))


@pytest.mark.parametrize("seed,n", [(1, 1), (2, 2), (3, 7), (4, 40), (5, 200), (6, 300)])
def test_bvh_builder_against_python_oracle_on_random_soups(rt, seed, n):
    from oracle import host_oracle as ho
    rng = np.random.RandomState(seed)
    centre = rng.uniform(-3, 3, (n, 1, 3))
    P = (centre + rng.uniform(-0.4, 0.4, (n, 3, 3))).astype(np.float32)
    if seed == 4:
        P[:, :, 1] = 0.25   # flat soup: one zero-extent axis is skipped (bvh.rs:329-331)
    if seed == 6:
        P = np.round(P, 1)  # coordinates on a grid: equal centroids, zeros of both signs (f32::min keeps self)
    v8 = np.zeros((n * 3, 8), np.float32)
    v8[:, :3] = P.reshape(-1, 3)
    v8[:, 3:6] = [0, 1, 0]
    sc = rt.Scene()
    sc.add_mesh_from_data(v8, np.arange(n * 3))
    sc.build()
    order, nodes = ho.build_bvh(P)
    tri, nd = sc.triangles(), sc.nodes()
    assert len(nd) == len(nodes)
    assert np.array_equal(bits(tri["v1"]), bits(P[order, 0]))
    assert nd["left"].tolist() == [x["left"] for x in nodes] and nd["right"].tolist() == [x["right"] for x in nodes]
    assert nd["first"].tolist() == [x["first"] for x in nodes] and nd["count"].tolist() == [x["count"] for x in nodes]
    assert np.array_equal(bits(nd["aabb_min"]), bits(np.array([x["mn"] for x in nodes], np.float32)))
    assert np.array_equal(bits(nd["aabb_max"]), bits(np.array([x["mx"] for x in nodes], np.float32)))
    # structural invariants: leaves partition the triangles, children boxes inside the parent
    leaves = nd[nd["count"] > 0]
    assert sorted(sum([list(range(f, f + c)) for f, c in zip(leaves["first"], leaves["count"])], [])) == list(range(n))
    for x in nd[nd["count"] == 0]:
        for c in (nd[x["left"]], nd[x["right"]]):
            assert np.all(c["aabb_min"] >= x["aabb_min"]) and np.all(c["aabb_max"] <= x["aabb_max"])


def test_bvh_quality_modes_and_empty_mesh(rt):
    v8 = np.zeros((6, 8), np.float32)
    v8[:, :3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 0, 0], [6, 0, 0], [5, 1, 0]]
    for q, n_nodes in ((1, 3), (0, 3), (2, 1)):   # High, Low split the two far triangles; Disabled keeps the root
        sc = rt.Scene()
        sc.add_mesh_from_data(v8, np.arange(6))
        sc.build(q)
        assert len(sc.nodes()) == n_nodes
    sc = rt.Scene()
    with pytest.raises(rt.RtError):
        sc.add_mesh_from_data(v8, [0, 1, 9])   # index out of range is a clean error


def test_glam_identities(rt):
    """Transform::cam for the Cornell camera; TRS matrix; inverse."""
    sc = rt.Scene()
    sc.set_camera((0, 1, 2), (0, 1, 0))
    c = np.array([list(col) for col in sc.uniform().camera.cam_to_world])
    assert np.array_equal(c, np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 1, 2, 1]], np.float32))
    # scale-rotation-translation: columns = (R x)*sx, (R y)*sy, (R z)*sz, (t, 1); 90 deg about Y
    h = np.float32(np.sqrt(0.5))
    v8 = np.zeros((3, 8), np.float32)
    v8[:, :3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    sc.add_mesh_from_data(v8, [0, 1, 2], xform=rt.transform(pos=(1, 2, 3), rot=(0, h, 0, h), scale=(2, 3, 4)))
    sc.build()
    m = sc.meshes()[0]
    m2w, w2m = m["model_to_world"].astype(np.float64), m["world_to_model"].astype(np.float64)
    assert np.allclose(m2w[0][:3], [0, 0, -2], atol=1e-6) and np.allclose(m2w[1][:3], [0, 3, 0], atol=1e-6)
    assert np.allclose(m2w[2][:3], [4, 0, 0], atol=1e-6) and np.allclose(m2w[3], [1, 2, 3, 1])
    assert np.allclose(w2m.T @ m2w.T, np.eye(4), atol=1e-6)   # column-major storage: rows here are columns


def test_camera_clamps_focus_distance(rt):
    sc = rt.Scene()
    sc.set_camera((0, 0, 3), (0, 0, -1), fov=45.0, focus_dist=0.1)   # Camera::new: focus_dist.max(1.0)
    vp = list(sc.uniform().camera.view_params)
    assert vp[2] == 1.0 and abs(vp[1] - 2 * np.tan(np.radians(22.5))) < 1e-6 and abs(vp[0] - vp[1] * 16 / 9) < 1e-6


def test_builtin_scene_library(rt):
    counts = {"room": (2, 6), "metal": (4, 0), "balls": (6, 0)}
    for name, (ns, nm) in counts.items():
        sc = rt.Scene.from_name(name, DATA)
        u = sc.uniform()
        assert (u.spheres, u.meshes) == (ns, nm), name
    with pytest.raises(rt.RtError):
        rt.Scene.from_name("no_such_scene", DATA)
    with pytest.raises(rt.RtError):
        rt.Scene.from_name("room_2", DATA)   # Dragon_80K.obj is absent: clean error, no abort


def test_random_balls_is_the_reference_construction_from_a_seeded_generator(rt):
    """scene.rs:365-444: floor + three big spheres + up to 22 x 22 small ones on a jittered grid, 80 % diffuse /
    15 % metal / 5 % glass, none closer than 0.9 to (4, 0.2, 0).  The reference seeds from the OS; here the
    seed is part of the name and the scene is reproducible."""
    a = rt.Scene.from_name("random_balls", DATA)
    b = rt.Scene.from_name("random_balls:0", DATA)
    c = rt.Scene.from_name("random_balls:7", DATA)
    sa, sb, sc = a.spheres(), b.spheres(), c.spheres()
    assert sa.tobytes() == sb.tobytes() and sa.tobytes() != sc.tobytes()
    assert 4 + 400 < len(sa) <= 4 + 484 <= 500           # ray_tracer.rs:16 caps spheres at 500
    assert sa["pos"][0].tolist() == [0, -1000, 0] and sa["radius"][0] == 1000
    small = sa[4:]
    assert np.all(small["radius"] == np.float32(0.2)) and np.all(small["pos"][:, 1] == np.float32(0.2))
    assert np.all(np.linalg.norm(small["pos"] - np.float32([4, 0.2, 0]), axis=1) > 0.9)
    gx, gz = np.floor(small["pos"][:, 0]), np.floor(small["pos"][:, 2])
    assert gx.min() >= -11 and gx.max() <= 10 and gz.min() >= -11 and gz.max() <= 10
    glass = small["material"]["flag"] == 1
    metal = (~glass) & (small["material"]["specular"] != np.float32(0.1))
    assert 0.01 < glass.mean() < 0.12 and 0.07 < metal.mean() < 0.25
    assert np.all(small["material"]["ior"][glass] == np.float32(1.3))
    assert np.all((small["material"]["color"][metal][:, :3] >= 0.5) & (small["material"]["color"][metal][:, :3] < 1.0))
    u = a.uniform()
    assert u.spheres == len(sa) and u.meshes == 0


def test_export_rgba8_against_the_oracle_restatement(rt, oracle):
    """rt_export_rgba8 == the literal restatement of app.rs:408-460 (reversed x loop + two flips) on a full
    golden frame and on the values the cast rules decide: NaN, +-inf, negatives, > 1, subnormals, values whose
    scaled result sits next to an integer.  Both sides call the platform's powf, as Rust's f32::powf does."""
    gold = np.load(os.path.join(GOLDEN, "cornell_golden.npz"))
    frames = [gold["frame_256_sky1"], gold["frame_256_sky0"]]
    rng = np.random.RandomState(9)
    odd = rng.uniform(-0.5, 1.5, (37, 53, 4)).astype(np.float32)   # odd sizes: the flips' middle row / column
    odd.reshape(-1)[:16] = [np.nan, np.inf, -np.inf, -0.0, 0.0, 1.0, 1e-45, -1e-45, 2.0, 0.99999994, 1.0000001, 0.5, 0.21404114, 1e30, -1.0, 3e-39]
    k = np.arange(256, dtype=np.float64)
    edges = ((k / 255.0) ** 2.2).astype(np.float32)                  # pow(v, 1/2.2) * 255 lands near the integer k
    edge_img = np.stack([np.nextafter(edges, np.float32(-1)), edges, np.nextafter(edges, np.float32(2)), edges], axis=1).reshape(16, 16, 4)
    L = rt.load()
    for img in frames + [odd, edge_img, np.zeros((1, 1, 4), np.float32), np.ones((2, 1, 4), np.float32)]:
        h, w = img.shape[:2]
        img = np.ascontiguousarray(img, np.float32)
        out = np.zeros((h, w, 4), np.uint8)
        assert L.rt_export_rgba8(img.ctypes.data, w, h, out.ctypes.data) == 0
        ref = oracle.export_rgba8(img)
        assert np.array_equal(out, ref)
        # independent of libm up to the truncation boundary: within one step of the double-precision formula
        with np.errstate(all="ignore"):
            d = np.clip(np.nan_to_num(np.power(img[::-1].astype(np.float64), 1 / 2.2), nan=0.0), 0, 1) * 255
        assert np.all(np.abs(out.astype(np.float64) - np.floor(d)) <= 1)


def test_export_rgba8_matches_save_render_to_file(rt):
    """app.rs:408-460: gamma 1/2.2, clamp, truncating u8, net vertical flip only."""
    img = np.zeros((2, 3, 4), np.float32)
    img[0, 0] = [1.0, 0.5, 0.0, 2.0]
    img[1, 2] = [0.25, np.nan, -1.0, 1.0]
    out = np.zeros((2, 3, 4), np.uint8)
    assert rt.load().rt_export_rgba8(img.ctypes.data, 3, 2, out.ctypes.data) == 0
    assert out[1, 0].tolist() == [255, int(0.5 ** (1 / 2.2) * 255), 0, 255]
    assert out[0, 2].tolist() == [int(np.float32(0.25) ** np.float32(1 / 2.2) * 255), 0, 0, 255]


def test_png_decoder_and_texture_flip(rt, tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    for mode, shape in (("RGB", (5, 7, 3)), ("RGBA", (4, 3, 4)), ("L", (6, 2)), ("P", (3, 3))):
        a = rng.randint(0, 256, shape, dtype=np.uint8)
        im = Image.fromarray(a, mode if mode != "P" else "L")
        if mode == "P":
            im = im.convert("P")
        im.save(tmp_path / "t.png")
        open(tmp_path / "t.mtl", "w").write("newmtl m\n Kd 1 1 1\n map_Kd t.png\n")
        open(tmp_path / "t.obj", "w").write("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl m\nf 1/1 2/2 3/3\n")
        sc = rt.Scene()
        sc.add_mesh_from_file("t.obj", assets_dir=str(tmp_path))
        tex = sc.textures()
        ref = np.array(Image.open(tmp_path / "t.png").convert("RGBA"))[:, ::-1]   # asset.rs:77 horizontal flip
        assert len(tex) == 1 and np.array_equal(tex[0], ref), mode
        sc.build()
        m = sc.meshes()["material"][0]
        assert m["flag"] == 2 and m["diffuse_index"] == 0 and m["normal_index"] == -1


def _built_arrays(rt, sc, **kw):
    sc.build(**kw)
    a = rt.SceneArrays.from_scene(sc)
    return a.nodes.tobytes(), a.triangles.tobytes(), a.meshes.tobytes()


@pytest.mark.parametrize("case", ["cornell", "dragon", "soup"])
def test_level_wise_bvh_build_is_identical(rt, case):
    """bvh_build_levels (breadth-first, searches in batches -- the host stand-in of the GPU search) must
    reproduce bvh_build: same nodes, same node numbering, same triangle order, bit for bit."""
    from ray_tracer_2_amd import scenes
    if case == "cornell":
        sc = scenes.cornell_from_raw(scenes.load_raw_meshes(os.path.join(GOLDEN, "cornell_raw.npz")))
    elif case == "dragon":
        sc = rt.Scene()
        for _label, v, idx, _t, _m in scenes.load_raw_meshes(os.path.join(GOLDEN, "dragon_raw.npz")):
            sc.add_mesh_from_data(v, idx)
    else:
        rng = np.random.default_rng(5)
        sc = rt.Scene()
        for n in (1, 2, 3, 17, 700):
            tri = rng.uniform(-1, 1, (n, 1, 3)) + rng.uniform(-0.2, 0.2, (n, 3, 3))
            tri[n // 2:] = np.round(tri[n // 2:], 1)   # ties: equal centroids, zero-extent axes
            v = np.concatenate([tri.reshape(-1, 3), np.tile([0, 1, 0, 0, 0], (3 * n, 1))], axis=1)
            sc.add_mesh_from_data(v, np.arange(3 * n))
    ref = _built_arrays(rt, sc)
    assert _built_arrays(rt, sc, device=-1, min_triangles=1) == ref


def test_device_bvh_build_fails_loudly_without_a_gpu(rt):
    """No silent fallback: asking for the GPU search on a machine without a HIP device is an error."""
    if rt.load().rt_device_count() > 0:
        pytest.skip("a GPU is present")
    from ray_tracer_2_amd import scenes
    sc = scenes.cornell_from_raw(scenes.load_raw_meshes(os.path.join(GOLDEN, "cornell_raw.npz")))
    with pytest.raises(rt.RtError) as e:
        sc.build(device=0, min_triangles=1)
    assert e.value.code == -3   # RT_ERR_DEVICE (include/rt_abi.h)


@pytest.mark.slow
def test_sponza_materials_and_loader_against_python_oracle(rt, tmp_path):
    """The reference's own sponza.mtl (tests/golden/sponza/sponza_standin.mtl: its 25 materials verbatim, fixture) and the OBJ
    files of the heterogeneous config 4 stand-in (scenes.sponza_hetero: procedural geometry, `vn` present in some files and left to
    the loader in others) through BOTH loaders: the C++ one the product uses (csrc/host/obj_loader.cpp, scene.cpp; tobj 4.0.3's
    model splitting, src/core/asset.rs:141-205 material rules, :224-261 normal synthesis) and the numpy restatement
    (oracle/host_oracle.py) -- materials of every group, vertices / normals / BVH of the groups small enough for the Python builder."""
    from oracle import host_oracle as ho
    from ray_tracer_2_amd import scenes
    sc = scenes.sponza_hetero(workdir=str(tmp_path))
    a = rt.SceneArrays.from_scene(sc)
    assert a.meshes.shape[0] == 393 and len(a.textures) == 25
    base = 0
    checked_geometry = 0
    for key in "ABCDE":
        models, mats, pos, tex, nrm = ho.load_obj(str(tmp_path / f"hetero_{key}.obj"))
        assert len(mats) == 25   # (every material of the MTL, used or not: tobj returns them all)
        for i, model in enumerate(models):
            got = a.meshes["material"][base + i]
            mat = ho.material_from_mtl(mats[model["material_id"]])
            for k, v in mat.items():
                if k in ("diffuse_index", "normal_index"):   # (texture slots are assigned by the loader that owns the array)
                    continue
                assert np.array_equal(np.float32(got[k]) if k != "flag" else got[k], np.float32(v) if k != "flag" else v), (key, i, k)
            # sponza.mtl: Kd 0.4704, Ks 0, Ns 7.843 on every material; a texture wherever map_Kd (or map_Disp) is given
            assert np.allclose(got["color"][:3], 0.4704) and got["specular"] == 0.0 and abs(got["smoothness"] - np.sqrt(np.float32(7.843137) / 100)) < 1e-6
            has_tex = "map_Kd" in mats[model["material_id"]] or "map_Disp" in mats[model["material_id"]].get("unknown", {})
            assert (got["flag"] == 2) == bool(has_tex), (key, i)
            if int(a.meshes["triangles"][base + i]) <= 160 and key in "CD" and checked_geometry < 40:
                P, N, UV = ho.unroll_model(model, pos, tex, nrm)
                order, nodes = ho.build_bvh(P)
                t0 = a.meshes["triangle_offset"][base + i]
                tri = a.triangles[t0:t0 + len(order)]
                assert np.array_equal(bits(tri["v1"]), bits(P[order, 0])) and np.array_equal(bits(tri["v3"]), bits(P[order, 2])), (key, i)
                assert np.array_equal(bits(tri["n2"]), bits(N[order, 1])), (key, i)   # (C: synthesised normals; D: the file's)
                n0 = a.meshes["node_offset"][base + i]
                nd = a.nodes[n0:n0 + len(nodes)]
                assert nd["count"].tolist() == [n["count"] for n in nodes] and nd["first"].tolist() == [n["first"] for n in nodes]
                assert np.array_equal(bits(nd["aabb_min"]), bits(np.array([n["mn"] for n in nodes], np.float32)))
                checked_geometry += 1
        base += len(models)
    assert base == 392 and checked_geometry >= 30   # (+ the emissive quad)
