"""Row-band sharding of one frame across ranks (one process per GPU).

Pixels are independent, so each rank renders a contiguous band of rows against its own full
copy of the mesh and the only exchange is the gather of the finished 8-bit bands to rank 0
(RCCL over xGMI when the process group is "nccl"; the same code runs on "gloo" for CPU tests).
"""
from __future__ import annotations

import numpy as np

from . import kernel


def band_rows(H: int, world: int, rank: int) -> tuple[int, int]:
    return kernel.band(H, world, rank)


def max_band_rows(H: int, world: int) -> int:
    return (H + world - 1) // world


def gather_bands(band, H: int, world: int, rank: int, dst: int = 0, out=None):
    """band: uint8 tensor [max_band_rows, W, 3] whose first (y1-y0) rows are this rank's band
    (bands are padded to a common height because gather needs equal shapes).
    Returns, on `dst`, the list of per-rank padded tensors; None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return [band]
    if rank == dst and out is None:
        out = [torch.empty_like(band) for _ in range(world)]
    dist.gather(band, out if rank == dst else None, dst=dst)
    return out if rank == dst else None


def assemble_rows(gathered, H: int, world: int):
    """Concatenates the valid rows of every rank's padded band: [H, W, 3] in y_pix order."""
    import torch

    parts = []
    for r in range(world):
        y0, y1 = band_rows(H, world, r)
        parts.append(gathered[r][: y1 - y0])
    return torch.cat(parts, 0)


def place_rows_host(rgb8_rows: np.ndarray) -> np.ndarray:
    """driver.c:186-190 on the host: pixel row y_pix goes to screen row H - y_pix, y_pix = 0
    falls off-screen, screen row 0 keeps the clear colour (driver.c:31-32)."""
    H = rgb8_rows.shape[0]
    screen = np.zeros_like(rgb8_rows)
    if H > 1:
        screen[1:] = rgb8_rows[1:][::-1]
    return screen
