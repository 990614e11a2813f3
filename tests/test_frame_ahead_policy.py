"""The automatic rule of option "frame_ahead" (csrc/rt_api.hip: ahead_depth), pinned on the CPU: how many frames a one-frame
call that continues an accumulation renders at once.  The figures are the ones include/rt_abi.h, INTEGRATION.md and
DESIGN.md section 6 quote; the measurements behind them: profiles/r04_strip_scaling.txt, r04_strip_batch_sweep.txt,
r04_frame_ahead_big_scenes.txt."""
import ray_tracer_2_amd as rt

W, H = 1920, 1080


def depth(lds, texels, spp=8, bounces=4, waits=0):
    return rt.load_test().rt_test_frame_ahead_depth(int(lds), texels, spp, bounces, int(waits))


def share(world):
    strips = (H + 7) // 8
    return (strips // world + (1 if strips % world else 0)) * 8 * W   # rank 0's share (rt_strip_texels)


def test_config2_shares_of_a_strip_split():
    # batches of about 4 ms of LDS-resident rays: 28 / 14 / 7 frames for a share of 8 / 4 / 2 ranks, none for the whole frame
    assert [depth(True, share(w)) for w in (8, 4, 2)] == [28, 14, 7]
    assert depth(True, W * H) == 0
    # a host that WAITS for every frame (its calls find the stream idle) never has a frame held back behind frames it has
    # not asked for (round 5): nothing is rendered ahead for it unless it sets an explicit depth
    assert depth(True, W * H, waits=1) == 0
    assert depth(True, share(8), waits=1) == 0
    # a small window: as many as a launch takes
    assert depth(True, 320 * 180) == 64


def test_scenes_read_from_global_memory():
    # rays of unknown cost: nothing is rendered ahead automatically (round 4 guessed "about 33 ms, at most 8 frames" from a work
    # estimate and needed a timing probe to take the guess back); a host that wants batches there sets frame_ahead itself
    assert depth(False, W * H) == 0 and depth(False, W * H, spp=16) == 0
    assert depth(False, 3840 * 2160, spp=64, bounces=8) == 0
    assert depth(False, share(8)) == 0 and depth(False, 320 * 180) == 0
    assert depth(False, W * H, waits=1) == 0


def test_nothing_to_render_ahead():
    assert depth(True, share(8), spp=0) == 0 and depth(False, share(8), spp=0) == 0
