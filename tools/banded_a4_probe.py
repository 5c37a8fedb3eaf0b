#!/usr/bin/env python3
"""Measures the cheapest CORRECT way of keeping conv_up1's output (a4) inside the 256 MB memory-side cache
between the two fused kernels: 1080p frames cut into horizontal bands with a 7-row halo (the same halo-tile
mechanism as sharding.py), so few bands per pass that a pass's a4 stays under a given size.  Prints the time per
1080p frame of the banded runs beside the whole-frame run of the same process.

  python tools/banded_a4_probe.py            # on an MI355X
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

W, H, HALO = 1920, 1080, 7


def timed(ctx, frames_equiv, reps=20):
    for _ in range(3):
        ctx.upscale(None)
    ctx.sync(None)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.upscale(None)
    ctx.sync(None)
    return (time.perf_counter() - t0) / reps / frames_equiv * 1e3


def main():
    import superresolution_amd as S
    rng = np.random.default_rng(1)
    with S.SuperResolutionContext(W, H, batch=8) as ctx:
        for i in range(8):
            f = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            f[..., 3] = 255
            ctx.upload_rgba8(i, f)
        print(f"whole frames, 8 per pass (a4 4.2 GB per pass):      {timed(ctx, 8):.3f} ms per 1080p frame")
    for bands, mb in [(8, 160), (8, 240), (4, 160), (4, 300), (2, 300)]:
        rows = H // bands + 2 * HALO                       # an inner band: halo on both sides
        os.environ["SRX_WORKSPACE_MB"] = str(mb)
        batch = 4 * bands                                  # four frames' worth of bands
        with S.SuperResolutionContext(W, rows, batch=batch) as ctx:
            for i in range(batch):
                f = rng.integers(0, 256, (rows, W, 4), dtype=np.uint8)
                f[..., 3] = 255
                ctx.upload_rgba8(i, f)
            per_pass = ctx.frames_per_launch
            a4_mb = per_pass * (2 * W) * (2 * rows) * 64 / 1e6
            print(f"{bands} bands of {rows} rows, {per_pass} bands per pass (a4 {a4_mb:.0f} MB per pass): "
                  f"{timed(ctx, 4):.3f} ms per 1080p frame")
    os.environ.pop("SRX_WORKSPACE_MB", None)


if __name__ == "__main__":
    main()
