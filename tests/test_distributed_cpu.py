"""world_size-2 test of the band sharding + gather path on the gloo backend (CPU)."""
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


def _worker(rank, world, port, W, H, full_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = np.load(full_path)                       # [H, W, 3] in y_pix order
        y0, y1 = parallel.band_rows(H, world, rank)
        band = torch.zeros((parallel.max_band_rows(H, world), W, 3), dtype=torch.uint8)
        band[: y1 - y0] = torch.from_numpy(full[y0:y1])  # what this rank "rendered"
        got = parallel.gather_bands(band, H, world, rank, dst=0)
        if rank == 0:
            img = parallel.assemble_rows(got, H, world).numpy()
            np.save(out_path, parallel.place_rows_host(img))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H", [(2, 37), (2, 48), (3, 50)])
def test_gather_of_row_bands(tmp_path, world, H):
    W = 40
    mesh = scenes.torus(150)
    fr = O.frame_setup(0)
    X, Y, Z = O.transform(mesh, O.arr4(fr.obmat))
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), fr, W, H, TAN_HALF)["rgb8"]
    full_path, out_path = str(tmp_path / "full.npy"), str(tmp_path / "out.npy")
    np.save(full_path, ref)
    mp.spawn(_worker, args=(world, _free_port(), W, H, full_path, out_path), nprocs=world, join=True)
    assert np.array_equal(np.load(out_path), O.place_pixels(ref))
