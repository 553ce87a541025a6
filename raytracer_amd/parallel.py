"""Row sharding of one frame across ranks (one process per GPU).

Pixels are independent, so each rank renders its rows against its own full copy of the mesh
and the only exchange is the gather of the finished 8-bit rows to rank 0 (RCCL over xGMI when
the process group is "nccl"; the same code runs on "gloo" for CPU tests).

Two ways to cut the image: contiguous bands (band_rows, the split of rt_band) and interleaved
parts (part_rows: blocks of 16 rows dealt round-robin).  The cost of a row is uneven -- the
shadow rays of a region cost what its occluders' positions in the triangle list dictate -- so
interleaving is what keeps the ranks level; bench.py uses it.
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


def part_rows(H: int, world: int, rank: int, block: int = 16):
    """Interleaved row set of `rank`: one block of `block` rows out of every `world`, dealt
    forwards or backwards after the Thue-Morse sequence (rt_rows in include/rt_amd.h)."""
    return kernel.rows_part(H, world, rank, block)


def max_part_rows(H: int, world: int, block: int = 16) -> int:
    return max(kernel.rows_count(part_rows(H, world, r, block)) for r in range(world))


def assemble_parts(gathered, H: int, world: int, block: int = 16):
    """Scatters the valid rows of every rank's padded part back into image order: [H, W, 3]."""
    import torch

    out = torch.empty((H,) + tuple(gathered[0].shape[1:]), dtype=gathered[0].dtype,
                      device=gathered[0].device)
    for r in range(world):
        rows = kernel.rows_image_rows(part_rows(H, world, r, block))
        out[torch.as_tensor(rows, device=out.device)] = gathered[r][: len(rows)]
    return out


def place_rows_host(rgb8_rows: np.ndarray) -> np.ndarray:
    """driver.c:186-190 on the host: pixel row y_pix goes to screen row H - y_pix, y_pix = 0
    falls off-screen, screen row 0 keeps the clear colour (driver.c:31-32)."""
    H = rgb8_rows.shape[0]
    screen = np.zeros_like(rgb8_rows)
    if H > 1:
        screen[1:] = rgb8_rows[1:][::-1]
    return screen
