"""Frame batches (rt_render_frames) and the multi-GPU entry points: the accumulated image after n
overlapped frames must be bit-identical to n sequential rt_render calls with Params.frames advancing
(app.rs:44-53 + wgsl:154-161), on one GPU, per strip, and through rt_render_multi."""
import os

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def sequential(rt, tracer, w, h, bounces, spp, f0, n, start=None):
    tracer.set_option("frame_ahead", 0)   # (one launch per frame: the plain path is the yardstick)
    try:
        tracer.write_image(np.zeros((h, w, 4), np.float32) if start is None else start)
        for f in range(f0, f0 + n):
            tracer.render(rt.make_params(w, h, bounces, spp, skybox=1, frames=f))
        return tracer.read_image(w, h)
    finally:
        tracer.set_option("frame_ahead", -1)


@pytest.fixture(scope="module")
def sponza(rt):
    from ray_tracer_2_amd import scenes
    return rt.SceneArrays.from_scene(scenes.sponza_standin(200))


@pytest.mark.parametrize("f0,n,batch", [(0, 8, 16), (0, 2, 16), (3, 20, 16), (1, 8, 3), (-1, 5, 16), (0, 33, 32), (2, 70, 64), (5, 1, 16)])
def test_render_frames_equals_sequential_frames(rt, tracer, cornell, f0, n, batch):
    w, h = 200, 100
    tracer.load_scene(cornell)
    start = np.random.RandomState(f0 + n).rand(h, w, 4).astype(np.float32)   # an earlier accumulation
    want = sequential(rt, tracer, w, h, 4, 4, f0, n, start)
    tracer.set_option("batch_frames", batch)
    try:
        tracer.write_image(start)
        tracer.reset_timing()
        tracer.render_frames(rt.make_params(w, h, 4, 4, skybox=1, frames=f0), n)
        got = tracer.read_image(w, h)
        st = tracer.stats()
    finally:
        tracer.set_option("batch_frames", 32)
    assert np.array_equal(bits(got), bits(want))
    assert st.frames == n and st.launches == -(-n // batch) and st.paths == w * h * 4 * n   # equal batches


def test_render_frames_against_the_oracle(rt, oracle, tracer, cornell):
    w, h = 96, 54
    tracer.load_scene(cornell)
    ref = np.zeros((h, w, 4), np.float32)
    segs = 0
    for f in range(5):
        ref, st = oracle.render(rt.make_params(w, h, 4, 8, skybox=1, frames=f), cornell, image=ref)
        segs += st.segments
    tracer.reset_timing()
    tracer.render_frames(rt.make_params(w, h, 4, 8, skybox=1, frames=0), 5)
    assert np.array_equal(bits(tracer.read_image(w, h)), bits(ref))
    assert tracer.stats().segments == segs


def test_render_frames_with_the_stats_counters(rt, oracle, tracer, cornell):
    """The counter kernels (wgsl:307,322 stats) in batch mode: the sums over the frames equal the oracle's."""
    w, h, n = 80, 48, 4
    tracer.load_scene(cornell)
    ref = np.zeros((h, w, 4), np.float32)
    seg = nt = tt = 0
    for f in range(n):
        ref, st = oracle.render(rt.make_params(w, h, 3, 2, skybox=1, frames=f), cornell, image=ref)
        seg, nt, tt = seg + st.segments, nt + st.node_tests, tt + st.triangle_tests
    tracer.set_counters(True)
    try:
        tracer.reset_timing()
        tracer.render_frames(rt.make_params(w, h, 3, 2, skybox=1, frames=0), n)
        got = tracer.read_image(w, h)
        s = tracer.stats()
    finally:
        tracer.set_counters(False)
    assert np.array_equal(bits(got), bits(ref))
    assert (s.segments, s.node_tests, s.triangle_tests) == (seg, nt, tt)
    assert s.segments_reused == 0   # the counter kernels re-intersect memoised rays so that the counters stay the shader's


@pytest.mark.parametrize("variant,lds,batch", [(0, 1, 0), (1, 1, 0), (0, 0, 0), (1, 0, 0), (0, 1, 6), (0, 0, 6)])
def test_segments_reused_is_exact(rt, tracer, cornell, variant, lds, batch):
    """rt_stats.segments_reused (segments served from the primary-ray memo): with the zero-strength Cornell camera every
    pixel's primary ray is constant (no -0 involved at an even width) and its hit is in the primary table (option
    primary_hits: computed once per camera, size and scene), so EVERY primary segment takes its hit from the memo: exactly
    W * H * spp per frame, whichever kernel renders the frame and however many lanes of a wave sit out an iteration.
    Without the hits in the table (primary_hits = 0) a pixel's first sample of every frame traverses: W * H * (spp - 1).
    The ray count is the same either way."""
    w, h, spp, frames = 200, 104, 5, 6
    tracer.load_scene(cornell)
    tracer.set_option("kernel_variant", variant)
    tracer.set_option("lds_scene", lds)
    tracer.set_option("frame_ahead", 0)   # (the counters count what was launched: exactly these frames, one launch each)
    got = {}
    try:
        for hits in (1, 0):
            tracer.set_option("primary_hits", hits)
            tracer.reset_timing()
            if batch:
                tracer.set_option("batch_frames", batch)
                tracer.render_frames(rt.make_params(w, h, 4, spp, skybox=1, frames=0), frames)
            else:
                for f in range(frames):
                    tracer.render(rt.make_params(w, h, 4, spp, skybox=1, frames=f))
            got[hits] = (tracer.stats(), tracer.read_image(w, h).copy())
    finally:
        tracer.set_option("kernel_variant", -1)
        tracer.set_option("lds_scene", 1)
        tracer.set_option("batch_frames", 32)
        tracer.set_option("primary_hits", 1)
        tracer.set_option("frame_ahead", -1)
    s, s0 = got[1][0], got[0][0]
    assert s.segments_reused == w * h * spp * frames and s0.segments_reused == w * h * (spp - 1) * frames
    assert s.segments == s0.segments and np.array_equal(bits(got[1][1]), bits(got[0][1]))
    assert s.paths == w * h * spp * frames and s0.segments - s0.segments_reused >= w * h * frames


def test_primary_table_follows_camera_scene_and_strip_layout(rt, oracle, cornell):
    """The primary table (every pixel's constant primary ray and its hit) is a function of camera, frame size, strip
    layout and scene: each of them changing between frames -- with pipelined frames in flight -- must rebuild it; the
    frames equal the oracle's throughout.  (A stale table would show the old camera's or the old scene's first hits.)"""
    w, h = 136, 80
    t = rt.RayTracer(0, w, h)
    try:
        t.set_option("pipeline_when_idle", 1)
        t.load_scene(cornell)

        def check(arrays, frames, width=w, height=h):
            ref = np.zeros((height, width, 4), np.float32)
            for f in range(frames):
                p = rt.make_params(width, height, 3, 2, skybox=1, frames=f)
                ref, _ = oracle.render(p, arrays, image=ref)
                t.render(p)
            assert np.array_equal(bits(t.read_image(width, height)), bits(ref))
        check(cornell, 3)
        cam_t = type(cornell.uniform.camera)
        moved = rt.SceneArrays(cornell.uniform, cornell.spheres, cornell.meshes, cornell.triangles, cornell.nodes)
        moved.uniform = type(cornell.uniform).from_buffer_copy(bytes(cornell.uniform))
        moved.uniform.camera.cam_to_world[3][0] += 0.21
        t.set_camera(cam_t.from_buffer_copy(bytes(moved.uniform.camera)))      # camera: new rays, new hits
        check(moved, 2)
        tris = cornell.triangles.copy()                                        # scene: same rays, new hits
        tris["v1"][:, 1] += 0.05
        tris["v2"][:, 1] += 0.05
        tris["v3"][:, 1] += 0.05
        lifted = rt.SceneArrays(moved.uniform, cornell.spheres, cornell.meshes, tris, cornell.nodes)
        nodes = cornell.nodes.copy()
        nodes["aabb_min"][:, 1] += 0.05
        nodes["aabb_max"][:, 1] += 0.05
        lifted = rt.SceneArrays(moved.uniform, cornell.spheres, cornell.meshes, tris, nodes)
        t.update_buffers(lifted)
        check(lifted, 2)
        check(lifted, 2, 96, 56)                                               # frame size
        # strip layout: rank 1 of 3 after the full frame, against the rows of the full frame
        full = t.read_image(96, 56).copy()
        t.write_image(np.zeros((h, w, 4), np.float32))
        for f in range(2):
            t.render_strips(rt.make_params(96, 56, 3, 2, skybox=1, frames=f), 1, 3)
        cnt = t.strip_texels(96, 56, 1, 3)
        mine = t.read_texels(cnt).reshape(-1, 96, 4)
        rows = [s * 8 + r for s in range(7) if s % 3 == 1 for r in range(8)]
        assert np.array_equal(bits(mine[:len(rows)]), bits(full[rows]))
    finally:
        t.close()


@pytest.mark.parametrize("kw", [dict(lds_scene=0), dict(pixel_cache=0), dict(pixel_cache=2), dict(tile_feedback=0), dict(batch_tile_major=0)])
def test_render_frames_under_every_kernel_option(rt, tracer, cornell, kw):
    w, h = 160, 90
    tracer.load_scene(cornell)
    want = sequential(rt, tracer, w, h, 4, 8, 0, 12)
    (k, v), = kw.items()
    tracer.set_option(k, v)
    try:
        tracer.write_image(np.zeros((h, w, 4), np.float32))
        # two batches of 6: the second one runs with the tile order the first one's costs gave
        tracer.set_option("batch_frames", 6)
        tracer.render_frames(rt.make_params(w, h, 4, 8, skybox=1, frames=0), 12)
        got = tracer.read_image(w, h)
    finally:
        tracer.set_option(k, {"lds_scene": 1, "pixel_cache": 1, "tile_feedback": 1, "batch_tile_major": 1}[k])
        tracer.set_option("batch_frames", 32)
    assert np.array_equal(bits(got), bits(want))


def test_render_frames_many_mesh_textured_scene(rt, tracer, sponza):
    """config 4's shape (many textured meshes under one transform, top-level tree kernels)."""
    w, h = 192, 108
    tracer.load_scene(sponza)
    want = sequential(rt, tracer, w, h, 3, 2, 0, 6)
    tracer.write_image(np.zeros((h, w, 4), np.float32))
    tracer.render_frames(rt.make_params(w, h, 3, 2, skybox=1, frames=0), 6)
    assert np.array_equal(bits(tracer.read_image(w, h)), bits(want))


def test_debug_views_and_empty_batches_take_the_single_frame_path(rt, tracer, cornell):
    w, h = 64, 36
    tracer.load_scene(cornell)
    for f in range(3):   # (a debug view is blended like any frame, wgsl:154-161)
        tracer.render(rt.make_params(w, h, 4, 1, debug_flag=2, debug_scale=8, frames=f))
    want = tracer.read_image(w, h)
    tracer.write_image(np.zeros((h, w, 4), np.float32))
    tracer.render_frames(rt.make_params(w, h, 4, 1, debug_flag=2, debug_scale=8, frames=0), 3)
    assert np.array_equal(bits(tracer.read_image(w, h)), bits(want))
    tracer.render_frames(rt.make_params(w, h, 4, 1), 0)   # nothing happens
    assert np.array_equal(bits(tracer.read_image(w, h)), bits(want))


@pytest.mark.parametrize("scene_name", ["cornell", "sponza"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_strip_frames_assemble_to_the_full_frames(rt, tracer, cornell, sponza, scene_name, world):
    """rt_render_strips_frames per rank + rt_assemble_strips == rt_render over the same frames."""
    arrays = cornell if scene_name == "cornell" else sponza
    w, h, spp, nb, n = 200, 100, 2, 3, 5   # 13 strips, ragged last strip
    tracer.load_scene(arrays)
    full = sequential(rt, tracer, w, h, nb, spp, 0, n)
    pad = tracer.strip_texels(w, h, 0, world)
    gathered = np.zeros((world, pad, 4), np.float32)
    small = rt.RayTracer(0, w, h)
    small.load_scene(arrays)
    for r in range(world):
        small.write_image(np.zeros((h, w, 4), np.float32))
        small.render_strips_frames(rt.make_params(w, h, nb, spp, skybox=1, frames=0), n, r, world)
        cnt = small.strip_texels(w, h, r, world)
        gathered[r, :cnt] = small.read_texels(cnt)
    stage = rt.RayTracer(0, world * pad, 1)
    stage.write_image(gathered.reshape(1, world * pad, 4))
    tracer.assemble_strips(stage.device_image_ptr, w, h, world)
    assert np.array_equal(bits(tracer.read_image(w, h)), bits(full))
    small.close()
    stage.close()


def test_raw_strip_buffers_agree_between_the_launch_paths(rt, cornell):
    """A rank's raw strip buffer -- the padding rows of a ragged last strip included (height % 8 != 0: they are what the
    gather ships, the assembled frame never reads them) -- is the same whether its frames were rendered in place
    (pipeline off), through the pipeline's scratch images or as one batch, also right after frames of another shape
    went through the same scratch images."""
    w, h, spp, nb, n, world = 200, 100, 2, 3, 4, 3   # 13 strips: rank 0 owns the ragged last one (rows 96..99 of 104)
    t = rt.RayTracer(0, 256, 128)
    try:
        t.load_scene(cornell)
        cnt = t.strip_texels(w, h, 0, world)
        got = {}
        for mode in ("plain", "pipeline", "batch"):
            t.set_option("pipeline", 0 if mode == "plain" else 4)
            t.set_option("pipeline_when_idle", 1)
            if mode == "pipeline":   # dirty the scratch images with full frames of another shape first
                for f in range(4):
                    t.render(rt.make_params(256, 128, 2, 1, skybox=1, frames=f))
            t.write_image(np.zeros((128, 256, 4), np.float32))
            if mode == "batch":
                t.render_strips_frames(rt.make_params(w, h, nb, spp, skybox=1, frames=0), n, 0, world)
            else:
                for f in range(n):
                    t.render_strips(rt.make_params(w, h, nb, spp, skybox=1, frames=f), 0, world)
            got[mode] = t.read_texels(cnt).copy()
        assert np.array_equal(bits(got["plain"]), bits(got["pipeline"]))
        assert np.array_equal(bits(got["plain"]), bits(got["batch"]))
        # the padding rows stay what the host wrote there (zeros): nobody samples them
        assert not got["plain"].reshape(-1, w, 4)[-4:].any()
    finally:
        t.close()


@pytest.mark.parametrize("batch", [1, 4, "rccl"])
def test_bench_rank_plumbing_on_torch_memory_and_stream(batch):
    """What `bench.py --gpus N` does per rank, all ranks in one process: every rank renders its strips straight into a
    torch tensor (rt_bind_image) on torch's current side stream (rt_set_stream), the tensors are stacked as the gather
    would deliver them, the root assembles into a torch frame (rt_assemble_strips) -- with batches and with one
    (pipelined) launch per frame, no host synchronisation inside a step; == rt_render over the same frames.
    (A process of its own, torch imported first as in bench.py: torch brings its own HIP runtime.)"""
    import subprocess
    import sys
    args = ["4", "rccl"] if batch == "rccl" else [str(batch)]   # "rccl": the gather through torch.distributed's nccl backend (one rank)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bench_plumbing.py"), *args],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "plumbing ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("scene_name", ["cornell", "sponza"])
def test_render_multi_without_readback_between_frames(rt, tracer, cornell, sponza, scene_name):
    """Several rt_render_multi calls back to back in the non-blocking mode (no host readback in
    between), one read at the end: every rank's next render has to wait for the root's copy of its
    previous strips (each handle has its own stream here, all on device 0)."""
    arrays = cornell if scene_name == "cornell" else sponza
    w, h, spp, nb = 320, 200, 2, 3
    tracer.load_scene(arrays)
    want = sequential(rt, tracer, w, h, nb, spp, 0, 6)
    ranks = [rt.RayTracer(0, w, h) for _ in range(3)]
    try:
        for t in ranks:
            t.load_scene(arrays)
        for f in range(6):
            rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=f), read_back=False)
        got = rt.read_multi_frame(ranks[0], w, h)
        assert np.array_equal(bits(got), bits(want))
        # and the same six frames as two overlapped batches of three
        for t in ranks:
            t.write_image(np.zeros((h, w, 4), np.float32))
        rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=0), read_back=False, n_frames=3)
        got = rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=3), read_back=True, n_frames=3)
        assert np.array_equal(bits(got), bits(want))
    finally:
        for t in ranks:
            t.close()


def test_render_multi_gather_through_rccl(rt, tracer, cornell):
    """The RCCL transport (grouped ncclSend/ncclRecv) on the one device this box has: a one-rank
    communicator gathering from itself.  On a multi-GPU node the same code runs with one rank per device."""
    w, h = 160, 90
    tracer.load_scene(cornell)
    want = sequential(rt, tracer, w, h, 4, 4, 0, 3)
    t = rt.RayTracer(0, w, h)
    try:
        t.load_scene(cornell)
        t.set_option("multi_rccl", 2)
        for f in range(2):
            rt.render_multi([t], rt.make_params(w, h, 4, 4, skybox=1, frames=f), read_back=False)
        got = rt.render_multi([t], rt.make_params(w, h, 4, 4, skybox=1, frames=2), read_back=True)
        assert np.array_equal(bits(got), bits(want))
    finally:
        t.close()


def test_render_multi_eight_handles_at_the_node_shape(rt, tracer, cornell):
    """rt_render_multi(_frames) with the 8 ranks of one node, all on this one device: a 1080-row frame is 135 strips
    dealt 17 x 7 + 16 (rank 7's buffer is a strip shorter than its padded gather slot).  Everything of the multi-GPU
    path runs here except the transfer between two devices itself (device-to-device copies on one device; the RCCL
    transport is covered with a one-rank communicator above and refuses handles that share a device): strip renders on
    eight streams, the root's copy events that the ranks' next renders wait for, gather layout, assemble -- frame by
    frame without read-back, in overlapped batches, with a rank's image read and written in between, and with the
    root destroyed first."""
    w, h, spp, nb = 72, 1080, 2, 3
    tracer.load_scene(cornell)
    want = sequential(rt, tracer, w, h, nb, spp, 0, 6)
    ranks = [rt.RayTracer(0, w, h) for _ in range(8)]
    try:
        for t in ranks:
            t.load_scene(cornell)
        assert [t.strip_texels(w, h, r, 8) // (8 * w) for r, t in enumerate(ranks)] == [17] * 7 + [16]
        for f in range(6):
            rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=f), read_back=False)
        assert np.array_equal(bits(rt.read_multi_frame(ranks[0], w, h)), bits(want))
        # a rank's image rewritten right behind a non-blocking call (the root may still be copying it), then batches
        for t in ranks:
            t.write_image(np.zeros((h, w, 4), np.float32))
        rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=0), read_back=False, n_frames=3)
        part = ranks[7].read_texels(ranks[7].strip_texels(w, h, 7, 8))      # rank 7's compact strips after 3 frames
        got = rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=3), read_back=True, n_frames=3)
        assert np.array_equal(bits(got), bits(want))
        three = sequential(rt, tracer, w, h, nb, spp, 0, 3)
        from ray_tracer_2_amd import parallel
        assert np.array_equal(bits(part.reshape(-1, w, 4)), bits(three[parallel.local_rows(h, 7, 8)]))
        # RCCL cannot hold two ranks of one device: insisting on it is an error, not a silent switch of transport
        ranks[0].set_option("multi_rccl", 2)
        with pytest.raises(rt.RtError):
            rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=0), read_back=False)
        ranks[0].set_option("multi_rccl", 1)
        # the root goes first, with copies of the others' strips possibly still in flight
        rt.render_multi(ranks, rt.make_params(w, h, nb, spp, skybox=1, frames=6), read_back=False)
        ranks[0].close()
        ranks[1].render(rt.make_params(w, h, nb, spp, skybox=1, frames=0))
        ranks[1].synchronize()
    finally:
        for t in ranks:
            t.close()


@pytest.mark.skipif("__import__('torch').cuda.device_count() < 2")
def test_render_multi_across_devices(rt, cornell):
    """Only on a multi-GPU node: one handle per device, RCCL gather and (multi_rccl = 0) peer copies."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    w, h = 320, 200
    single = rt.RayTracer(0, w, h)
    single.load_scene(cornell)
    want = sequential(rt, single, w, h, 4, 4, 0, 4)
    single.close()
    for transport in (1, 0):
        ranks = [rt.RayTracer(d, w, h) for d in range(n)]
        try:
            for t in ranks:
                t.load_scene(cornell)
            ranks[0].set_option("multi_rccl", transport)
            for f in range(3):
                rt.render_multi(ranks, rt.make_params(w, h, 4, 4, skybox=1, frames=f), read_back=False)
            got = rt.render_multi(ranks, rt.make_params(w, h, 4, 4, skybox=1, frames=3), read_back=True)
            assert np.array_equal(bits(got), bits(want))
        finally:
            for t in ranks:
                t.close()


def test_timing_survives_the_event_pool_wrap(rt, tracer, cornell):
    """More launches than the event pool holds: launches, frames, paths and kernel_ms keep counting."""
    w, h = 8, 8
    tracer.load_scene(cornell)
    tracer.set_option("frame_ahead", 0)   # (one launch per frame)
    try:
        tracer.reset_timing()
        n = 4096 + 50
        for f in range(n):
            tracer.render(rt.make_params(w, h, 1, 1, frames=f))
        st = tracer.stats()
    finally:
        tracer.set_option("frame_ahead", -1)
    assert st.launches == n and st.frames == n and st.paths == 64 * n
    assert st.kernel_ms > 0 and st.kernel_ms / n < 5.0


def test_independent_frames_pipelined_across_two_handles(rt, oracle, cornell):
    """Two handles on one device (each its own stream and image) rendering independent frames alternately, without
    a synchronisation in between -- the recipe for a moving camera: every frame equals the
    oracle's."""
    W, H = 160, 90
    hs = [rt.RayTracer(0, W, H) for _ in range(2)]
    try:
        for h in hs:
            h.load_scene(cornell)
        refs = {}
        for f in (0, 3, 7, 12):   # (frames = 0 stores; frames >= 1 would blend with the handle's own previous image)
            p = rt.make_params(W, H, 4, 4, skybox=1, frames=0)
            p.frames = -f   # the seed is |frames| * 719393; frames <= 0: no blend (wgsl:154-161)
            refs[f] = oracle.render(p, cornell)[0]
        order = [0, 3, 7, 12]
        for i, f in enumerate(order):      # all four launches in flight before anything is read
            p = rt.make_params(W, H, 4, 4, skybox=1, frames=0)
            p.frames = -f
            hs[i & 1].render(p)
        # the last frame of each handle is what its image holds
        assert np.array_equal(hs[0].read_image(W, H).view(np.uint32), refs[7].view(np.uint32))
        assert np.array_equal(hs[1].read_image(W, H).view(np.uint32), refs[12].view(np.uint32))
        # and read in between: frame by frame, alternating
        for i, f in enumerate(order):
            p = rt.make_params(W, H, 4, 4, skybox=1, frames=0)
            p.frames = -f
            hs[i & 1].render(p)
            hs[(i + 1) & 1].render(p)      # the other handle renders the same frame concurrently
            assert np.array_equal(hs[i & 1].read_image(W, H).view(np.uint32), refs[f].view(np.uint32)), f
            assert np.array_equal(hs[(i + 1) & 1].read_image(W, H).view(np.uint32), refs[f].view(np.uint32)), f
    finally:
        for h in hs:
            h.close()


def test_pipelined_single_frames(rt, oracle, tracer, cornell):
    """Option pipeline (default: four frames in flight): consecutive rt_render calls sample into scratch images on internal
    streams and are blended in frame order on the handle's stream.  Forty frames without a synchronisation in between -- across tile-order
    refreshes (every 8 frames), a camera change (primary-ray table rebuilt), a change of frame size, a batch in the middle,
    an image written by the host -- equal the same calls with the pipeline off, and the first frames equal the oracle's;
    every frame is observable: reading after any call returns that frame."""
    w, h = 200, 104

    def script(t):
        t.write_image(np.zeros((h, w, 4), np.float32))
        out = []
        for f in range(20):
            t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))
            if f in (0, 1, 2, 9):
                out.append(t.read_image(w, h).copy())          # (observable at any point)
        cam = type(cornell.uniform.camera).from_buffer_copy(bytes(cornell.uniform.camera))
        cam.cam_to_world[3][0] += 0.05
        t.set_camera(cam)                                       # accumulation restarts: frames = 0 stores
        for f in range(12):
            t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))
        t.render_frames(rt.make_params(w, h, 4, 3, skybox=1, frames=12), 5)   # a batch on the handle's stream in between
        for f in range(17, 25):
            t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))
        out.append(t.read_image(w, h).copy())
        for f in range(6):                                      # another frame size: the scratch images are re-made
            t.render(rt.make_params(96, 54, 3, 2, skybox=1, frames=f))
        out.append(t.read_image(96, 54).copy())
        t.set_camera(cornell.uniform.camera)
        return out

    tracer.load_scene(cornell)
    try:
        tracer.set_option("frame_ahead", 0)   # (every frame a launch of its own: this test is about those)
        tracer.set_option("pipeline", 0)
        want = script(tracer)
        # (pipeline_when_idle = 1: frames this small are over before the next call arrives, and a frame that finds the
        # stream idle takes the plain launch by default -- here every frame goes through the pipeline; the last run is
        # the default rule, whichever way each frame goes)
        for depth, when_idle in ((2, 1), (3, 1), (4, 1), (4, 0)):   # frames in flight
            tracer.set_option("pipeline", depth)
            tracer.set_option("pipeline_when_idle", when_idle)
            got = script(tracer)
            assert len(got) == len(want)
            for k, (g, wnt) in enumerate(zip(got, want)):
                assert np.array_equal(bits(g), bits(wnt)), (depth, when_idle, k)
    finally:
        tracer.set_option("pipeline", 1)
        tracer.set_option("pipeline_when_idle", 0)
        tracer.set_option("frame_ahead", -1)
    ref = np.zeros((h, w, 4), np.float32)
    for f in range(3):
        ref, _ = oracle.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f), cornell, image=ref)
        assert np.array_equal(bits(got[f]), bits(ref)), f


def test_frames_rendered_ahead_leave_the_same_image_after_every_call(rt, oracle, tracer, cornell):
    """Option frame_ahead: a one-frame call that continues an accumulation renders the next frames with its own in one
    batched launch and the following calls only blend theirs.  The image after EVERY call -- through a host that re-sends
    the same camera each frame (the reference's update_buffers), a change of samples per pixel, a moved camera
    (accumulation restarts), a skipped frame number, an image written by the host, a batch and a strip call in between,
    a sequence longer than a batch -- equals the one-launch-per-frame run's, and the first frames equal the oracle's."""
    w, h = 200, 104
    cam_t = type(cornell.uniform.camera)

    def script(t):
        out = []

        def frame(f, spp=3, ww=w, hh=h):
            t.render(rt.make_params(ww, hh, 4, spp, skybox=1, frames=f))
            out.append(t.read_image(ww, hh).copy())
        t.write_image(np.zeros((h, w, 4), np.float32))
        for f in range(12):
            if f % 3 == 0:
                t.set_camera(cam_t.from_buffer_copy(bytes(cornell.uniform.camera)))   # the same camera again
            frame(f)
        frame(12, spp=2)                                         # other parameters: what was rendered ahead is dropped
        for f in range(13, 16):
            frame(f, spp=2)
        cam = cam_t.from_buffer_copy(bytes(cornell.uniform.camera))
        cam.cam_to_world[3][0] += 0.05
        t.set_camera(cam)                                        # accumulation restarts
        for f in range(5):
            frame(f)
        frame(7)                                                 # a frame number skipped
        frame(8)
        t.write_image(out[3])                                    # the host restores an earlier image
        for f in range(9, 12):
            frame(f)
        t.render_frames(rt.make_params(w, h, 4, 3, skybox=1, frames=12), 5)   # an explicit batch in between
        for f in range(17, 20):
            frame(f)
        t.render_strips(rt.make_params(w, h, 4, 3, skybox=1, frames=20), 1, 4)   # (a strip share overwrites the image's head)
        out.append(t.read_texels(t.strip_texels(w, h, 1, 4)).copy())
        for f in range(40):                                      # longer than any batch; no read in between
            t.render(rt.make_params(96, 54, 3, 2, skybox=1, frames=f))
        out.append(t.read_image(96, 54).copy())
        t.set_camera(cornell.uniform.camera)
        return out

    tracer.load_scene(cornell)
    try:
        tracer.set_option("frame_ahead", 0)
        want = script(tracer)
        for ahead in (-1, 2, 5, 32, 64):
            tracer.set_option("frame_ahead", ahead)
            tracer.reset_timing()
            got = script(tracer)
            st = tracer.stats()
            assert len(got) == len(want)
            for k, (g, wnt) in enumerate(zip(got, want)):
                assert np.array_equal(bits(g), bits(wnt)), (ahead, k)
            if ahead > 0:
                assert st.launches < st.frames   # (frames were rendered ahead: fewer launches than frames)
            else:   # automatic: this script reads -- waits for -- most frames; nothing is rendered ahead for a host that waits
                assert st.launches <= st.frames
    finally:
        tracer.set_option("frame_ahead", -1)
    ref = np.zeros((h, w, 4), np.float32)
    for f in range(3):
        ref, _ = oracle.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f), cornell, image=ref)
        assert np.array_equal(bits(want[f]), bits(ref)), f


def test_pipelined_single_frames_global_memory_scene(rt, tracer):
    """The same on a scene read from global memory (its primary-ray memo lives in global memory, one buffer per
    concurrent launch) and on a many-mesh textured scene: 12 frames, pipeline on == off."""
    from ray_tracer_2_amd import scenes
    w, h = 160, 90
    for arrays in (rt.SceneArrays.from_scene(scenes.sponza_standin(340, detail=8)), rt.SceneArrays.from_scene(scenes.sponza_standin(200))):
        tracer.load_scene(arrays)
        outs = []
        try:
            tracer.set_option("pipeline_when_idle", 1)   # (every frame through the pipeline, see above)
            for pipe in (0, 3, 4):
                tracer.set_option("pipeline", pipe)
                tracer.write_image(np.zeros((h, w, 4), np.float32))
                tracer.reset_timing()
                for f in range(12):
                    tracer.render(rt.make_params(w, h, 3, 2, skybox=1, frames=f))
                outs.append((tracer.read_image(w, h).copy(), tracer.stats().segments))
        finally:
            tracer.set_option("pipeline", 1)
            tracer.set_option("pipeline_when_idle", 0)
        for o in outs[1:]:
            assert np.array_equal(bits(outs[0][0]), bits(o[0])) and outs[0][1] == o[1]


def test_burst_after_an_idle_frame_with_the_memo_in_global_memory(rt, tracer, cornell):
    """pixel_cache = 2 (the per-wave primary-ray memo in global memory, stateful across a pixel's samples): a frame that
    finds the stream idle takes the plain launch on the handle's stream, and the pipelined frames behind it do not wait
    for it -- so they must not share its memo (slot 0 used to).  Bursts of frames big enough to overlap,
    a synchronisation before each burst, pipeline on == off bit for bit."""
    w, h = 960, 544
    tracer.load_scene(cornell)
    outs = []
    try:
        tracer.set_option("frame_ahead", 0)
        tracer.set_option("pixel_cache", 2)
        for pipe in (0, 4, 2):
            tracer.set_option("pipeline", pipe)
            tracer.write_image(np.zeros((h, w, 4), np.float32))
            got = []
            f = 0
            for burst in (2, 3, 5, 2):
                tracer.synchronize()             # the next call finds the stream idle
                for _ in range(burst):
                    tracer.render(rt.make_params(w, h, 4, 6, skybox=1, frames=f))
                    f += 1
                got.append(tracer.read_image(w, h).copy())
            outs.append(got)
    finally:
        tracer.set_option("pipeline", 1)
        tracer.set_option("pixel_cache", 1)
        tracer.set_option("frame_ahead", -1)
    for o in outs[1:]:
        for k, (g, wnt) in enumerate(zip(o, outs[0])):
            assert np.array_equal(bits(g), bits(wnt)), k


def test_snapshot_is_the_frame_of_its_call_while_later_frames_render(rt, cornell):
    """rt_snapshot_image / rt_read_snapshot: the snapshot taken behind frame k is frame k -- bit for bit -- although
    frames k + 1.. were queued before it was read (the reference's host shows every frame: src/rendering/renderer.rs),
    under pipelined launches, frames rendered ahead and with both off; it can be read twice and in part."""
    w, h, n = 320, 180, 9
    t = rt.RayTracer(0, w, h)
    with pytest.raises(rt.RtError):
        t.read_snapshot(w, h)                                   # nothing taken yet
    t.load_scene(cornell)
    with pytest.raises(rt.RtError):
        t.snapshot_image(w + 1, h)                              # larger than the image
    t.set_option("frame_ahead", 0)
    t.set_option("pipeline", 0)
    want = []
    for f in range(n):
        t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))
        want.append(t.read_image(w, h).copy())
    for ahead, pipe in ((0, 0), (0, 4), (-1, 1), (4, 1)):
        t.set_option("frame_ahead", ahead)
        t.set_option("pipeline", pipe)
        t.write_image(np.zeros((h, w, 4), np.float32))
        got = []
        t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=0))
        t.snapshot_image(w, h)
        for f in range(1, n):
            t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))   # queued before frame f - 1 is read
            got.append(t.read_snapshot(w, h))
            if f == 3:
                assert np.array_equal(bits(t.read_snapshot(w, h)), bits(got[-1]))                 # twice
                assert np.array_equal(bits(t.read_snapshot(w, h // 2)), bits(got[-1][: h // 2]))   # in part
            t.snapshot_image(w, h)
        got.append(t.read_snapshot(w, h))
        for f in range(n):
            assert np.array_equal(bits(got[f]), bits(want[f])), (ahead, pipe, f)
        assert np.array_equal(bits(t.read_image(w, h)), bits(want[-1]))
    t.close()


def test_moving_camera_frames_overlap_on_slot_tables(rt, oracle, cornell):
    """Option primary_per_slot: a pipelined frame whose camera is not the shared primary table's builds a table of its own
    pipeline slot (no barrier), so the frames of a moving camera are in flight together.  Every frame of a camera that
    moves each call, then stands (accumulation on the slots' tables), then moves back -- taken with rt_snapshot_image
    while later frames are queued -- equals the run without pipeline and the oracle's."""
    w, h = 200, 104
    cam_t = type(cornell.uniform.camera)

    def cam_at(dx):
        c = cam_t.from_buffer_copy(bytes(cornell.uniform.camera))
        c.cam_to_world[3][0] += dx
        return c

    moves = [(0.01 * k, 0) for k in range(1, 9)] + [(0.08, f) for f in range(1, 7)] + [(0.0, 0), (0.0, 1), (0.0, 2)]

    def script(t):
        out = []
        t.write_image(np.zeros((h, w, 4), np.float32))
        for k, (dx, f) in enumerate(moves):
            t.set_camera(cam_at(dx))
            t.render(rt.make_params(w, h, 4, 3, skybox=1, frames=f))
            if k:
                out.append(t.read_snapshot(w, h))      # the frame before, while this one is queued
            t.snapshot_image(w, h)
        out.append(t.read_snapshot(w, h))
        t.set_camera(cornell.uniform.camera)
        return out

    t = rt.RayTracer(0, w, h)
    t.load_scene(cornell)
    t.set_option("frame_ahead", 0)
    t.set_option("pipeline", 0)
    want = script(t)
    t.set_option("pipeline_when_idle", 1)   # (frames this small finish before the next call: force them through the pipeline)
    for pipe, per_slot in ((4, 1), (3, 1), (4, 0), (8, 1)):
        t.set_option("pipeline", pipe)
        t.set_option("primary_per_slot", per_slot)
        got = script(t)
        for k, (g, wnt) in enumerate(zip(got, want)):
            assert np.array_equal(bits(g), bits(wnt)), (pipe, per_slot, k)
    t.close()
    # the oracle on the first moved frame
    a1 = rt.SceneArrays(cornell.uniform, cornell.spheres, cornell.meshes, cornell.triangles, cornell.nodes)
    a1.uniform = type(cornell.uniform).from_buffer_copy(bytes(cornell.uniform))
    a1.uniform.camera.cam_to_world[3][0] += 0.01
    ref, _ = oracle.render(rt.make_params(w, h, 4, 3, skybox=1, frames=0), a1)
    assert np.array_equal(bits(want[0]), bits(ref))
