"""Round 5: the boundary's safe defaults.

 * a host that renders, WAITS and renders again is never handed its frame behind frames it has not asked for (automatic
   `frame_ahead` only batches for a host that runs ahead of the device); rt_get_stats counts the frames asked for and
   the frames rendered ahead separately;
 * option `max_device_mb` bounds what the library allocates on its own initiative -- scratch images, tables, memos, park
   queues -- and what does not fit takes the plainer path: same image, bit for bit; rt_last_launch reports what is held;
 * a pipeline deeper than the hardware queues the host asked the runtime for is said once through rt_last_error."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_a_waiting_host_gets_one_launch_per_frame_and_exact_counters(rt, oracle, tracer, cornell):
    w, h, nf = 320, 184, 6
    tracer.load_scene(cornell)
    tracer.write_image(np.zeros((h, w, 4), np.float32))
    tracer.reset_timing()
    ref = np.zeros((h, w, 4), np.float32)
    want = 0
    for f in range(nf):
        p = rt.make_params(w, h, 4, 4, skybox=1, frames=f)
        ref, st = oracle.render(p, cornell, image=ref)
        want += st.segments
        tracer.render(p)
        assert np.array_equal(bits(tracer.read_image(w, h)), bits(ref)), f   # (the read waits for the frame)
    s = tracer.stats()
    assert (s.frames, s.frames_speculative, s.launches, s.segments) == (nf, 0, nf, want)


def test_frames_rendered_ahead_are_counted_apart(rt, tracer, cornell):
    w, h = 320, 184
    tracer.load_scene(cornell)
    try:
        tracer.set_option("frame_ahead", 8)   # the host opts in: batches of 8 whatever it does
        tracer.write_image(np.zeros((h, w, 4), np.float32))
        tracer.reset_timing()
        for f in range(11):                   # frame 0 alone, a batch for 1..8, a batch for 9..16 of which 2 are asked for
            tracer.render(rt.make_params(w, h, 4, 2, skybox=1, frames=f))
        s = tracer.stats()
        assert s.frames == 11 and s.frames_speculative == 6 and s.launches == 3
        a = tracer.read_image(w, h).copy()
        tracer.render(rt.make_params(w, h, 4, 3, skybox=1, frames=11))   # other parameters: the rest of the batch is dropped
        s = tracer.stats()
        assert s.frames == 12 and s.frames_speculative == 6 and s.launches == 4
        tracer.set_option("frame_ahead", 0)
        tracer.write_image(np.zeros((h, w, 4), np.float32))
        for f in range(11):
            tracer.render(rt.make_params(w, h, 4, 2, skybox=1, frames=f))
        assert np.array_equal(bits(tracer.read_image(w, h)), bits(a))
    finally:
        tracer.set_option("frame_ahead", -1)


@pytest.mark.parametrize("cap_mb", [0, 200, 40, 1])
def test_max_device_mb_bounds_the_optional_allocations(rt, cornell, cap_mb):
    """1280 x 720: a scratch image is 14.7 MB, a primary table 59 MB.  200 MB: room for a table and some batch frames; 40 MB:
    no table, two scratch frames; 1 MB: plain launches only.  The image is the same every time."""
    w, h, nf = 1280, 720, 12
    t = rt.RayTracer(device=0, max_width=w, max_height=h)
    try:
        t.set_option("max_device_mb", cap_mb)
        t.load_scene(cornell)
        t.render_frames(rt.make_params(w, h, 3, 2, skybox=1, frames=0), nf)
        for f in range(nf, nf + 6):           # single-frame calls behind it: pipeline scratch images, slot tables
            t.render(rt.make_params(w, h, 3, 2, skybox=1, frames=f))
        img = t.read_image(w, h).copy()
        ll = t.last_launch()
        assert ll["device_mb_cap"] == cap_mb
        if cap_mb:
            assert ll["device_mb_held"] <= cap_mb, ll
        else:
            assert ll["device_mb_held"] >= 12 * 14   # (12 scratch frames at least)
    finally:
        t.close()
    want = getattr(test_max_device_mb_bounds_the_optional_allocations, "_img", None)
    if want is None:
        test_max_device_mb_bounds_the_optional_allocations._img = img
    else:
        assert np.array_equal(bits(img), bits(want)), cap_mb


def test_a_pipeline_deeper_than_the_hardware_queues_is_said_once(rt, cornell):
    import os
    queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    t = rt.RayTracer(device=0, max_width=256, max_height=144)
    try:
        t.load_scene(cornell)
        t.set_option("pipeline", 8)           # nine streams: more than any queue count the suite runs with (<= 8)
        t.set_option("pipeline_when_idle", 1)
        L = rt.load()
        for f in range(3):
            t.render(rt.make_params(256, 144, 2, 1, skybox=1, frames=f))
            msg = L.rt_last_error(t._h).decode()
            if f == 0 and queues < 9:
                assert msg.startswith("note: GPU_MAX_HW_QUEUES is") and "pipeline = 8" in msg, msg
        t.synchronize()
    finally:
        t.close()


def test_max_device_mb_switches_the_deferred_walks_off_not_the_image(rt):
    """A big-mesh scene whose batches defer the mesh's walks (two park queues: 2 x 144 B per pixel and frame): under a cap that
    leaves no room for the queues the same frames come from the plain kernels -- another frame time, the same bits."""
    import os
    from conftest import ROOT
    from ray_tracer_2_amd import scenes
    g = os.path.join(ROOT, "tests", "golden")
    a = rt.SceneArrays.from_scene(scenes.cornell_dragon(scenes.load_raw_meshes(os.path.join(g, "cornell_raw.npz")),
                                                        scenes.load_raw_meshes(os.path.join(g, "dragon_raw.npz")), subdivide=3))
    w, h, nf = 1920, 1080, 8   # (8 units of work: the automatic rule defers from there; the queues would take 7.4 GB)
    imgs, deferred = [], []
    for cap in (0, 300):
        t = rt.RayTracer(device=0, max_width=w, max_height=h)
        try:
            t.set_option("max_device_mb", cap)
            t.set_option("batch_frames", nf)
            t.load_scene(a)
            t.render_frames(rt.make_params(w, h, 4, 16, skybox=1, frames=0), nf)
            imgs.append(t.read_image(w, h).copy())
            ll = t.last_launch()
            deferred.append(ll["deferred_walks"])
            if cap:
                assert ll["device_mb_held"] <= cap, ll
        finally:
            t.close()
    assert deferred == [True, False], deferred
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
