"""Image-space partition across ranks and the framebuffer gather.

Rows are grouped into blocks of `rows_per_block`; rank r of N owns blocks r, r+N, ...
(interleaved so sky / ground cost spreads evenly).  Each rank renders its owned rows
packed into a [max_local_rows, W, C] tensor; the only exchange step is one gather of
those tensors to rank 0 (RCCL over xGMI on GPUs, gloo in CPU tests) followed by a
de-interleave into the full framebuffer.  Pixels are independent (per-pixel RNG
streams keyed by x + y*W, rng.cuh:12-14), so the result is identical for any N.

This mirrors, in torch, the row arithmetic of mort_hip_local_rows / mort_hip_global_row.
"""
import torch


def num_blocks(height, rows_per_block):
    return (height + rows_per_block - 1) // rows_per_block


def local_rows(height, rank, nranks, rows_per_block):
    n = 0
    for b in range(rank, num_blocks(height, rows_per_block), nranks):
        r0 = b * rows_per_block
        n += min(r0 + rows_per_block, height) - r0
    return n


def max_local_rows(height, nranks, rows_per_block):
    return ((num_blocks(height, rows_per_block) + nranks - 1) // nranks) * rows_per_block


def global_rows(rank, nranks, rows_per_block, n_local, device=None):
    """Global row index of each of the first n_local packed rows of `rank` (may run past the image for padding rows)."""
    l = torch.arange(n_local, device=device)
    return ((l // rows_per_block) * nranks + rank) * rows_per_block + (l % rows_per_block)


class FrameGather:
    """Pre-computed index maps + buffers for gathering packed row tiles to rank 0."""

    def __init__(self, height, width, channels, dtype, rank, world_size, rows_per_block, device):
        self.h, self.rank, self.n = height, rank, world_size
        self.max_lr = max_local_rows(height, world_size, rows_per_block)
        self.tile = torch.zeros((self.max_lr, width, channels), dtype=dtype, device=device)
        self.frame = torch.zeros((height, width, channels), dtype=dtype, device=device) if rank == 0 else None
        self.gathered = None
        self.maps = None
        if world_size > 1 and rank == 0:
            self.gathered = [torch.empty_like(self.tile) for _ in range(world_size)]
            self.maps = []
            for r in range(world_size):
                g = global_rows(r, world_size, rows_per_block, self.max_lr, device)
                valid = g < height
                self.maps.append((valid, g[valid]))

    def gather(self, dist):
        """tile (this rank's packed rows) -> frame on rank 0.  With one rank the tile is the frame."""
        if self.n == 1:
            self.frame[: self.h].copy_(self.tile[: self.h])
            return self.frame
        dist.gather(self.tile, self.gathered if self.rank == 0 else None, dst=0)
        if self.rank == 0:
            for r in range(self.n):
                valid, g = self.maps[r]
                self.frame.index_copy_(0, g, self.gathered[r][valid])
        return self.frame
