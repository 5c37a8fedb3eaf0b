"""Partition logic (CPU) and its two-process form over gloo: frames and halo tiles cover the work
exactly once, no exchange step is needed, and the job time is the slowest rank's."""
import os
import socket
import sys

import numpy as np
import pytest

from superresolution_amd import sharding
from conftest import ROOT


def test_frames_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(f for r in range(world) for f in sharding.frames_for_rank(64, r, world))
        assert seen == list(range(64))
        assert {len(sharding.frames_for_rank(64, r, world)) for r in range(world)} == {64 // world}


@pytest.mark.parametrize("w,h,tx,ty", [(7680, 4320, 1, 8), (7680, 4320, 2, 4), (96, 80, 2, 2), (91, 91, 3, 2), (10, 10, 4, 4)])
def test_tile_plan_covers_image(w, h, tx, ty):
    tiles = sharding.tile_plan(w, h, tx, ty)
    cover = np.zeros((h, w), np.int32)
    for t in tiles:
        cover[t.y0:t.y1, t.x0:t.x1] += 1
        assert t.ex0 == max(0, t.x0 - 7) and t.ex1 == min(w, t.x1 + 7)
        assert t.ey0 == max(0, t.y0 - 7) and t.ey1 == min(h, t.y1 + 7)
        # cropping the (2x) output of the extended rectangle gives exactly the interior
        fake = np.zeros((2 * (t.ey1 - t.ey0), 2 * (t.ex1 - t.ex0), 4), np.uint8)
        assert t.crop(fake).shape[:2] == (2 * (t.y1 - t.y0), 2 * (t.x1 - t.x0))
    assert (cover == 1).all()
    for world in (1, 2, 8):
        got = [t for r in range(world) for t in sharding.tiles_for_rank(tiles, r, world)]
        assert sorted(got, key=lambda t: (t.y0, t.x0)) == sorted(tiles, key=lambda t: (t.y0, t.x0))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from superresolution_amd import sharding as sh
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = sh.frames_for_rank(10, rank, world)
    tiles = sh.tiles_for_rank(sh.tile_plan(96, 80, 2, 2), rank, world)
    # every rank works alone; afterwards the shards are only compared, never exchanged on the data path
    mine = torch.zeros(10 + 4, dtype=torch.int64)
    for f in frames:
        mine[f] += 1
    plan = sh.tile_plan(96, 80, 2, 2)
    for t in tiles:
        mine[10 + plan.index(t)] += 1
    dist.all_reduce(mine)
    # the optional gather of tile interiors to one rank
    import numpy as np
    W, H = 50, 38
    truth = np.random.default_rng(5).integers(0, 256, (2 * H, 2 * W, 4), dtype=np.uint8)
    big = sh.tile_plan(W, H, 3, 2)
    interiors = [np.ascontiguousarray(truth[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1]) for t in sh.tiles_for_rank(big, rank, world)]
    whole = sh.gather_tiles(interiors, W, H, 3, 2, rank, world, dst=1)
    assert (whole is None) == (rank != 1)
    if rank == 1:
        assert np.array_equal(whole, truth)
    slowest = sh.max_over_ranks(1.0 + rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine.tolist(), slowest))


def test_two_ranks_gloo():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, counts, slowest in results:
        assert counts == [1] * 14          # every frame and every tile exactly once across ranks
        assert slowest == 2.0              # the job takes as long as its slowest rank


def test_gather_tiles_single_rank():
    W, H = 21, 17
    truth = np.random.default_rng(6).integers(0, 256, (2 * H, 2 * W, 4), dtype=np.uint8)
    plan = sharding.tile_plan(W, H, 2, 3)
    interiors = [np.ascontiguousarray(truth[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1]) for t in plan]
    assert np.array_equal(sharding.gather_tiles(interiors, W, H, 2, 3, 0, 1), truth)
    with pytest.raises(ValueError):
        sharding.gather_tiles(interiors[:-1], W, H, 2, 3, 0, 1)
