"""N > 1 path on CPU: world_size-2/3/8 `gloo` processes split the frame into
8-row strips, render their strips (with the CPU oracle standing in for the HIP
library), do ONE gather, and rank 0's assembled frame must be bit-identical to
the single-process frame -- the multi-GPU parity property of SURVEY.md 8e."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import ray_tracer_2_amd as rt
    from oracle import oracle
    from ray_tracer_2_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    arrays = rt.SceneArrays.load(os.path.join(GOLDEN, "cornell_scene.npz"))
    pad = parallel.pad_texels(width, height, world)
    local = torch.zeros((pad, 4), dtype=torch.float32)
    # frame 1 goes through the form bench.py uses: a preallocated receive buffer and a caller-supplied assembly
    gathered = torch.empty((world, pad, 4), dtype=torch.float32) if rank == 0 else None
    frame = None
    full = np.zeros((height, width, 4), np.float32)       # this rank's scratch full-frame target
    for f in range(2):                                     # frame 1 exercises per-rank accumulation
        p = rt.make_params(width, height, 2, 2, skybox=1, frames=f)
        rows = parallel.local_rows(height, rank, world)
        oracle.render(p, arrays, image=full, rows=np.array(rows, np.uint32), threads=2)
        buf = local.numpy().reshape(-1, width, 4)
        buf[:len(rows)] = full[rows]                       # compact strip-major layout (8 rows per strip)
        if f == 0:
            frame = parallel.gather_frame(dist, local, width, height, rank, world)
        else:
            frame = parallel.gather_frame(dist, local, width, height, rank, world, gathered=gathered,
                                          assemble_fn=lambda g, w, h, n: parallel.assemble(g, w, h, n))
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


# (8, 16 x 1080): the strip counts of the 8-GPU node at 1080 rows -- 135 strips dealt 17 x 7 + 16, so the last rank's
# buffer is one strip shorter than the padded gather slot
@pytest.mark.parametrize("world,shape", [(2, (64, 40)), (3, (48, 72)), (8, (16, 1080))])
def test_strip_split_gather_is_bit_identical(rt, oracle, cornell, tmp_path, world, shape):
    import torch.multiprocessing as mp
    width, height = shape
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    got = np.load(out)
    ref = np.zeros((height, width, 4), np.float32)
    for f in range(2):
        ref, _ = oracle.render(rt.make_params(width, height, 2, 2, skybox=1, frames=f), cornell, image=ref)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_strip_math_matches_library(rt):
    from ray_tracer_2_amd import parallel
    L = rt.load()
    for (w, h) in [(1920, 1080), (64, 36), (7, 9), (8, 8), (100, 1)]:
        for world in (1, 2, 3, 8):
            total = 0
            for r in range(world):
                n = parallel.local_texels(w, h, r, world)
                assert n == L.rt_strip_texels(w, h, r, world)
                total += len(parallel.local_rows(h, r, world))
            assert total == h
    # 1080 rows = 135 strips over 8 GPUs -> 17 x 7 + 16
    assert [len(parallel.local_strips(1080, r, 8)) for r in range(8)] == [17] * 7 + [16]


def test_assemble_inverts_the_split(rt):
    from ray_tracer_2_amd import parallel
    w, h, world = 5, 21, 4
    frame = np.arange(h * w * 4, dtype=np.float32).reshape(h, w, 4)
    pad = parallel.pad_texels(w, h, world)
    g = np.zeros((world, pad, 4), np.float32)
    for r in range(world):
        rows = parallel.local_rows(h, r, world)
        g[r].reshape(-1, w, 4)[:len(rows)] = frame[rows]
    assert np.array_equal(parallel.assemble(g, w, h, world), frame)
