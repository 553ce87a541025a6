"""world_size-2/3 tests of the row sharding + gather path on the gloo backend (CPU): contiguous
bands and interleaved parts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from raytracer_amd import parallel, scenes
from conftest import TAN_HALF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, full_path, out_path, block):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = np.load(full_path)                       # [H, W, 3] in y_pix order
        if block:       # interleaved parts
            from raytracer_amd import kernel
            mine = kernel.rows_image_rows(parallel.part_rows(H, world, rank, block))
            band = torch.zeros((parallel.max_part_rows(H, world, block), W, 3), dtype=torch.uint8)
            band[: len(mine)] = torch.from_numpy(full[mine])
        else:
            y0, y1 = parallel.band_rows(H, world, rank)
            band = torch.zeros((parallel.max_band_rows(H, world), W, 3), dtype=torch.uint8)
            band[: y1 - y0] = torch.from_numpy(full[y0:y1])  # what this rank "rendered"
        got = parallel.gather_bands(band, H, world, rank, dst=0)
        if rank == 0:
            img = (parallel.assemble_parts(got, H, world, block) if block
                   else parallel.assemble_rows(got, H, world)).numpy()
            np.save(out_path, parallel.place_rows_host(img))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,block", [(2, 37, 0), (2, 48, 0), (3, 50, 0), (2, 37, 16), (3, 50, 4),
                                           (2, 10, 16)])
def test_gather_of_rows(tmp_path, world, H, block):
    W = 40
    mesh = scenes.torus(150)
    fr = O.frame_setup(0)
    X, Y, Z = O.transform(mesh, O.arr4(fr.obmat))
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), fr, W, H, TAN_HALF)["rgb8"]
    full_path, out_path = str(tmp_path / "full.npy"), str(tmp_path / "out.npy")
    np.save(full_path, ref)
    mp.spawn(_worker, args=(world, _free_port(), W, H, full_path, out_path, block), nprocs=world, join=True)
    assert np.array_equal(np.load(out_path), O.place_pixels(ref))
