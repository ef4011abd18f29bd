"""Stripe sharding and the rank-0 gather (rayhs_amd/tiles.py), on CPU with gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rayhs_amd import tiles


@pytest.mark.parametrize("world,height,stripe", [(1, 17, 8), (2, 1080, 8), (8, 1080, 8), (8, 2160, 16), (3, 10, 4), (4, 3, 8)])
def test_stripes_partition_the_rows(world, height, stripe):
    seen = []
    for r in range(world):
        rows = tiles.stripe_rows_of(r, world, height, stripe)
        assert rows == sorted(rows)
        assert all(((y // stripe) % world) == r for y in rows)
        seen += rows
    assert sorted(seen) == list(range(height))
    assert tiles.max_rows(world, height, stripe) >= (height + world - 1) // world


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, width, stripe, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = tiles.FrameGather(rank, world, height, width, stripe, device="cpu")
        for frame_no in range(2):  # plan is reusable frame after frame
            g.local.zero_()
            for i, y in enumerate(g.rows):  # stand-in for the kernel: row y holds y, x, frame
                g.local[i, :, 0] = float(y)
                g.local[i, :, 1] = torch.arange(width, dtype=torch.float32)
                g.local[i, :, 2] = float(frame_no)
            frame = g.gather()
            if rank == 0:
                np.save(os.path.join(out_dir, f"frame{frame_no}.npy"), frame.numpy())
            else:
                assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,width,stripe", [(37, 5, 4), (16, 3, 8)])
def test_gather_assembles_the_frame_world_size_2(tmp_path, height, width, stripe):
    mp.spawn(_worker, args=(2, _free_port(), height, width, stripe, str(tmp_path)), nprocs=2, join=True)
    for frame_no in range(2):
        f = np.load(tmp_path / f"frame{frame_no}.npy")
        assert f.shape == (height, width, 3)
        assert np.array_equal(f[:, :, 0], np.repeat(np.arange(height, dtype=np.float32)[:, None], width, 1))
        assert np.array_equal(f[:, :, 1], np.repeat(np.arange(width, dtype=np.float32)[None, :], height, 0))
        assert np.all(f[:, :, 2] == frame_no)


def test_single_rank_plan_needs_no_process_group():
    g = tiles.FrameGather(0, 1, 9, 2, 4, device="cpu")
    g.local[:] = 3.0
    assert torch.all(g.gather() == 3.0) and g.gather().shape == (9, 2, 3)
