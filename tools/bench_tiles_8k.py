#!/usr/bin/env python3
"""The spatial partition at full size: one synthetic 7680x4320 image cut into eight horizontal bands with a
7-pixel halo, the tiles dealt to the ranks (one process per GPU, no exchange on the data path).

  python tools/bench_tiles_8k.py [--steps 10 --warmup 2]                      # one GPU: all eight tiles
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
      --master-port 29511 tools/bench_tiles_8k.py                             # one tile per GPU

Prints one JSON line: output megapixels/s for the whole image (slowest rank's time), tiles resident
in HBM when the clock starts.  With --gather the tile outputs are also collected on rank 0 (outside
the timed region) and compared with nothing: it only shows the optional last step working."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
W, H, TX, TY = 7680, 4320, 1, 8       # eight bands of 540 rows: full-width tiles keep the strips whole


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gather", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    import superresolution_amd as S
    from superresolution_amd import sharding

    rng = np.random.default_rng(4320)
    mine = sharding.tiles_for_rank(sharding.tile_plan(W, H, TX, TY), rank, world)
    ctxs = []
    for t in mine:          # every rank makes only its own tiles' pixels (seeded per tile)
        w, h = t.ex1 - t.ex0, t.ey1 - t.ey0
        tile_rng = np.random.default_rng(4320 + 131 * t.y0 + t.x0)
        img = tile_rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        ctx = S.SuperResolutionContext(w, h, device=local_rank)
        ctx.upload_rgba8(0, img)
        ctxs.append(ctx)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        for ctx in ctxs:
            ctx.upscale()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    barrier()
    if args.gather:
        interiors = [np.ascontiguousarray(t.crop(ctx.tensor_to_image(0))) for t, ctx in zip(mine, ctxs)]
        whole = sharding.gather_tiles(interiors, W, H, TX, TY, rank, world, dst=0, device="cuda")
        assert (whole is None) == (rank != 0)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        print(json.dumps({"metric": "output megapixels/sec (7680x4320 -> 15360x8640, 8 halo tiles)",
                          "value": round(4 * W * H / 1e6 / (ms / 1e3), 1), "unit": "output megapixels/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                          "tiles_per_gpu": len(mine), "halo": sharding.HALO, "data": "synthetic"}))
    for ctx in ctxs:
        ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
