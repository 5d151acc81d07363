"""Tile-row sharding of one frame over the GPUs of a node.

Every tile is independent (own RNG seed from its coordinates, disjoint output rectangle, read-only
scene), so rank r of N renders tile rows r, r+N, r+2N, ... (cyclic: the character sits in the
middle rows, contiguous bands would leave the edge ranks with pure background).  There is no
collective inside the render; one gather to rank 0 assembles the frame.  The same code drives RCCL
(backend "nccl", device tensors) and gloo (CPU tensors, used by the world_size-2 tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from .abi import Config


def tile_rows(cfg: Config) -> int:
    if cfg.width <= 0 or cfg.height <= 0 or cfg.tileSize <= 0:
        return 0
    return (cfg.height + cfg.tileSize - 1) // cfg.tileSize


def owned_tile_rows(cfg: Config, rank: int, world: int) -> List[int]:
    return list(range(rank, tile_rows(cfg), world))


def packed_rows(cfg: Config, world: int) -> int:
    """Pixel rows of every rank's packed buffer (padded so all ranks send the same count)."""
    return ((tile_rows(cfg) + world - 1) // world) * cfg.tileSize


def unpack_rows(cfg: Config, rank: int, world: int, packed: torch.Tensor, frame: torch.Tensor) -> None:
    """packed: (packed_rows, W, C) rows of `rank` in owned order → frame (H, W, C).  Works on any
    device; the GPU bench uses the HIP kernel (mcrt_unpack_rows_device) instead."""
    ts = cfg.tileSize
    for k, tr in enumerate(owned_tile_rows(cfg, rank, world)):
        y0 = tr * ts
        n = min(ts, cfg.height - y0)
        frame[y0:y0 + n] = packed[k * ts:k * ts + n]


def gather_frame(cfg: Config, packed: torch.Tensor, rank: int, world: int, frame: Optional[torch.Tensor] = None,
                 gather_bufs: Optional[List[torch.Tensor]] = None, async_op: bool = False):
    """One gather to rank 0.  Returns (work_or_None, gather_bufs).  The caller un-permutes with
    unpack_rows / the HIP unpack kernel once the work has completed."""
    if rank == 0 and gather_bufs is None:
        gather_bufs = [torch.empty_like(packed) for _ in range(world)]
    work = dist.gather(packed, gather_bufs if rank == 0 else None, dst=0, async_op=async_op)
    return work, gather_bufs
