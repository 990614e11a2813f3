"""Tile-split data parallelism across the GPUs of one node.

Pixels are independent (own RNG stream seeded by the full-frame pixel index,
own accumulator texel), so the frame shards with no exchange during rendering:
the frame is cut into 8-row strips (the wave tile height) dealt round-robin,
strip s -> rank s % world.  Each rank renders its strips into a compact local
buffer ([local_strips * 8, W, 4] f32) and keeps its own accumulation across
frames; ONE gather per displayed frame assembles the image on rank 0.

This module is the host-side glue over `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPUs; "gloo" in the CPU tests).  It never computes
pixels: `render_local` is the HIP library on a GPU and the CPU oracle in tests.
"""
import numpy as np


def n_strips(height):
    return (height + 7) // 8


def local_strips(height, rank, world):
    """Global strip indices owned by `rank`, in local order."""
    return list(range(rank, n_strips(height), world))


def local_rows(height, rank, world):
    rows = []
    for s in local_strips(height, rank, world):
        rows += list(range(s * 8, min(s * 8 + 8, height)))
    return rows


def local_texels(width, height, rank, world):
    """Texels of the compact local buffer (ragged last strip padded to 8 rows);
    equals rt_strip_texels()."""
    return len(local_strips(height, rank, world)) * 8 * width


def pad_texels(width, height, world):
    """Every rank's buffer is padded to rank 0's size so one equal-size gather works."""
    return local_texels(width, height, 0, world)


def assemble(gathered, width, height, world):
    """[world, pad_texels, 4] (torch tensor or ndarray) -> [height, width, 4].
    Reference implementation of rt_assemble_strips for CPU tests."""
    xp_full = gathered.new_zeros if hasattr(gathered, "new_zeros") else None
    out = xp_full((height, width, 4)) if xp_full else np.zeros((height, width, 4), gathered.dtype)
    for r in range(world):
        buf = gathered[r].reshape(-1, width, 4)
        for ls, s in enumerate(local_strips(height, r, world)):
            rows = min(8, height - s * 8)
            out[s * 8:s * 8 + rows] = buf[ls * 8:ls * 8 + rows]
    return out


def gather_frame(dist, local, width, height, rank, world, assemble_fn=None, gathered=None):
    """One collective per displayed frame (or per batch of accumulated frames): every rank contributes
    its padded local buffer, rank 0 receives [world, pad_texels, 4] and assembles the frame.
    `local` is a torch tensor of shape [pad_texels, 4] (device or CPU); `gathered` an optional
    preallocated receive buffer on rank 0; `assemble_fn(gathered, width, height, world)` replaces the
    reference assembly (bench.py passes rt_assemble_strips on the device)."""
    import torch
    pad = pad_texels(width, height, world)
    assert local.shape == (pad, 4)
    if rank == 0 and gathered is None:
        gathered = torch.empty((world, pad, 4), dtype=local.dtype, device=local.device)
    dist.gather(local, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    return (assemble_fn or assemble)(gathered, width, height, world)
