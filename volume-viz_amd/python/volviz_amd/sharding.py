"""Host logic of the multi-GPU screen-tile shard: which pixel rows a rank owns,
how a rank compacts them, and how rank 0 re-assembles the gathered frame.

The reference has no multi-GPU code; this is new design.  A frame is cut into bands of
`BAND_SLABS` slab rows (14 pixel rows each, kernel.cu:418) dealt round-robin to the ranks, so
every rank gets a similar share of long (centre) and short (edge) rays.  There is no exchange
during the march -- the volume is replicated -- and one gather of RGBA8 bands per frame.
Works on any torch device / backend (nccl = RCCL on the GPUs, gloo on CPU in the tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

SLAB = 14            # BLOCK_WIDTH - 2, kernel.cu:30,418
BAND_SLABS = 4       # slab rows per band (multiple of 4: 56 pixel rows = 7 strips of 8)
BAND_PX = SLAB * BAND_SLABS


def padded_height(height: int, world: int) -> int:
    """Frame buffers are allocated with a height that is a whole number of band rounds."""
    step = BAND_PX * world
    return (height + step - 1) // step * step


def owned_rows(height: int, world: int, rank: int):
    """Pixel rows y with ((y / 14) / BAND_SLABS) % world == rank -- the predicate of
    vv_render_options.shard_* (include/volviz.h)."""
    return [y for y in range(height) if ((y // SLAB) // BAND_SLABS) % world == rank]


def shard_option(world: int, rank: int):
    """(shard_band, shard_count, shard_index) for volviz_amd.make_options(shard=...)."""
    return (BAND_SLABS, world, rank) if world > 1 else None


def compact(frame_padded: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    """frame_padded: uint8 [H_pad, W, 4] -> this rank's bands, contiguous [nb, BAND_PX, W, 4]."""
    hp, w, c = frame_padded.shape
    return frame_padded.view(hp // BAND_PX, BAND_PX, w, c)[rank::world].contiguous()


def gather_frame(frame_padded: torch.Tensor, world: int, rank: int, recv=None, group=None):
    """One collective per frame: every rank sends its compacted bands to rank 0, which writes
    them back at their rows.  Returns the assembled padded frame on rank 0 (None elsewhere).
    `recv` (rank 0) may hold preallocated receive buffers to keep allocation out of the loop."""
    if world == 1:
        return frame_padded
    mine = compact(frame_padded, world, rank)
    if rank == 0:
        if recv is None:
            recv = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, gather_list=recv, dst=0, group=group)
        hp, w, c = frame_padded.shape
        out = frame_padded.view(hp // BAND_PX, BAND_PX, w, c)
        for r in range(1, world):
            out[r::world] = recv[r]
        return frame_padded
    dist.gather(mine, gather_list=None, dst=0, group=group)
    return None
