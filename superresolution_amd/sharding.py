"""Partitioning of the path across GPUs: by frame and by spatial tile with halo.

Frames and tiles are independent units (the network has no cross-image dependency: batch-norm
is folded into the filters), so partitioning needs no exchange step.  One process drives one
GPU; torch.distributed is used only by callers that want a common clock (bench.py)."""
from dataclasses import dataclass

import numpy as np

HALO = 7   # low-resolution pixels: 2+1+1 for conv1..3, then ceil((2+1+1+1)/2) = 3 for conv_up1..conv6


def frames_for_rank(total_frames, rank, world):
    """Frames {rank, rank + world, ...}: the frame-batch partition."""
    return list(range(rank, total_frames, world))


@dataclass(frozen=True)
class Tile:
    x0: int     # interior rectangle in input pixels [x0, x1) x [y0, y1)
    y0: int
    x1: int
    y1: int
    ex0: int    # rectangle actually fed to the network: interior + halo, clipped to the image
    ey0: int
    ex1: int
    ey1: int

    def crop(self, out):
        """Interior part (2x grid) of the network output for the extended rectangle."""
        return out[2 * (self.y0 - self.ey0):2 * (self.y1 - self.ey0), 2 * (self.x0 - self.ex0):2 * (self.x1 - self.ex0)]


def tile_plan(width, height, tiles_x, tiles_y, halo=HALO):
    """Split the image into tiles_x * tiles_y rectangles, each extended by `halo` real pixels
    where a neighbour exists.  At a true image border there is no halo: the network's own zero
    padding applies there, exactly as in the untiled run."""
    xs = [round(i * width / tiles_x) for i in range(tiles_x + 1)]
    ys = [round(j * height / tiles_y) for j in range(tiles_y + 1)]
    tiles = []
    for j in range(tiles_y):
        for i in range(tiles_x):
            x0, x1, y0, y1 = xs[i], xs[i + 1], ys[j], ys[j + 1]
            if x1 > x0 and y1 > y0:
                tiles.append(Tile(x0, y0, x1, y1, max(0, x0 - halo), max(0, y0 - halo),
                                  min(width, x1 + halo), min(height, y1 + halo)))
    return tiles


def tiles_for_rank(tiles, rank, world):
    return tiles[rank::world]


def upscale_tiled(rgba, tiles_x, tiles_y, device=0, halo=HALO, rank=0, world=1, out=None):
    """Run this rank's tiles of one image on `device` and paste their interiors into `out`
    (uint8 [2H, 2W, 4]; created if None).  With world == 1 the whole image is produced."""
    from . import SuperResolutionContext
    h, w = rgba.shape[:2]
    if out is None:
        out = np.zeros((2 * h, 2 * w, 4), np.uint8)
    ctxs = {}
    for t in tiles_for_rank(tile_plan(w, h, tiles_x, tiles_y, halo), rank, world):
        size = (t.ex1 - t.ex0, t.ey1 - t.ey0)
        if size not in ctxs:
            ctxs[size] = SuperResolutionContext(size[0], size[1], device=device)
        res = ctxs[size].run(np.ascontiguousarray(rgba[t.ey0:t.ey1, t.ex0:t.ex1]))
        out[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1] = t.crop(res)
    for c in ctxs.values():
        c.close()
    return out


def gather_tiles(interiors, width, height, tiles_x, tiles_y, rank, world, dst=0, halo=HALO, device="cpu"):
    """Optional last step of the tile partition: collect the ranks' tile interiors on rank `dst`
    (one torch.distributed gather: RCCL over xGMI with the nccl backend and device="cuda", gloo on
    the CPU).  `interiors` holds this rank's interiors (uint8 [2h, 2w, 4] each) in the order of
    tiles_for_rank.  Returns the assembled uint8 [2H, 2W, 4] image on `dst`, None elsewhere.
    Nothing on the data path needs this; it exists for consumers that want one buffer."""
    import torch
    import torch.distributed as dist
    plan = tile_plan(width, height, tiles_x, tiles_y, halo)
    per_rank = [tiles_for_rank(plan, r, world) for r in range(world)]
    nbytes = [sum(16 * (t.x1 - t.x0) * (t.y1 - t.y0) for t in ts) for ts in per_rank]
    if len(interiors) != len(per_rank[rank]):
        raise ValueError("gather_tiles: one interior per tile of this rank is required")
    flat = np.zeros(max(max(nbytes), 1), np.uint8)
    pos = 0
    for t, a in zip(per_rank[rank], interiors):
        if a.shape != (2 * (t.y1 - t.y0), 2 * (t.x1 - t.x0), 4) or a.dtype != np.uint8:
            raise ValueError("gather_tiles: interior of tile %r has the wrong shape or type" % (t,))
        flat[pos:pos + a.size] = a.reshape(-1)
        pos += a.size
    if not (dist.is_available() and dist.is_initialized()):
        if world != 1:
            raise RuntimeError("gather_tiles: world > 1 needs an initialised torch.distributed process group")
        parts = [torch.from_numpy(flat)]
    else:
        mine = torch.from_numpy(flat).to(device)
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, parts, dst=dst)
        if rank != dst:
            return None
    out = np.zeros((2 * height, 2 * width, 4), np.uint8)
    for r, ts in enumerate(per_rank):
        buf = parts[r].cpu().numpy()
        pos = 0
        for t in ts:
            n = 16 * (t.x1 - t.x0) * (t.y1 - t.y0)
            out[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1] = buf[pos:pos + n].reshape(2 * (t.y1 - t.y0), 2 * (t.x1 - t.x0), 4)
            pos += n
    return out


def max_over_ranks(seconds):
    """Slowest rank's time (what a frame-sharded job takes); identity without a process group.
    With a process group the reduction runs whatever the world size, so a one-rank job exercises
    the same calls as an eight-rank one."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value):
    """Every rank's `value` (a float), in rank order, on every rank; [value] without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]
