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


class FrameGatherer:
    """Double-buffered, asynchronous form of gather_frame for a stream of frames: the gather of
    frame k runs (on the collective's own stream, over xGMI) while frame k+1 is being rendered
    into the other buffer.  Per step:

        g.finish(b)          # frame that used buffer b two steps ago is complete on rank 0
        render into g.frames[b] (owned rows only)
        g.submit(b)          # send this rank's bands, do not wait

    and g.drain() after the last step.  Nothing is exchanged while a frame is marched; this only
    moves the one collective per frame off the critical path.  Buffers are allocated once."""

    def __init__(self, height: int, width: int, world: int, rank: int, device=None, depth: int = 2, group=None):
        self.world, self.rank, self.group = world, rank, group
        hp = padded_height(height, world)
        self.frames = [torch.zeros((hp, width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]
        nb = hp // BAND_PX // world
        self.mine = [torch.empty((nb, BAND_PX, width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]
        # rank 0 receives into one tensor per buffer, [world, nb, BAND_PX, W, 4], so that the bands of
        # all ranks go back to their rows with a single strided copy
        self.recv_all = [torch.empty((world, nb, BAND_PX, width, 4), dtype=torch.uint8, device=device)
                         if (rank == 0 and world > 1) else None for _ in range(depth)]
        self.recv = [[ra[r] for r in range(world)] if ra is not None else None for ra in self.recv_all]
        self.work = [None] * depth

    def submit(self, b: int) -> None:
        if self.world == 1:
            return
        assert self.work[b] is None, "finish(b) first"
        f = self.frames[b]
        hp, w, c = f.shape
        self.mine[b].copy_(f.view(hp // BAND_PX, BAND_PX, w, c)[self.rank::self.world])
        self.work[b] = dist.gather(self.mine[b], gather_list=self.recv[b], dst=0, group=self.group, async_op=True)

    def finish(self, b: int):
        """Waits for the gather submitted on buffer b (if any); on rank 0 writes the other ranks'
        bands back at their rows and returns the assembled padded frame."""
        if self.world == 1:
            return self.frames[b]
        if self.work[b] is None:
            return None
        self.work[b].wait()              # the current stream now orders after the collective
        self.work[b] = None
        if self.rank != 0:
            return None
        f = self.frames[b]
        hp, w, c = f.shape
        # band k*world + r of the frame is band k of rank r (rank 0's own bands come back unchanged)
        f.view(hp // BAND_PX // self.world, self.world, BAND_PX, w, c).copy_(self.recv_all[b].transpose(0, 1))
        return f

    def drain(self):
        return [self.finish(b) for b in range(len(self.frames))]
