#!/usr/bin/env python3
"""Time every stage of an upscale with HIP events (development helper).

  python tools/time_stages.py [--reps 30] [--w 1920 --h 1080] [--batch 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    import torch
    import superresolution_amd as S
    rng = np.random.default_rng(0)
    ctx = S.SuperResolutionContext(a.w, a.h, batch=a.batch, flags=S.FLAG_KEEP_TENSORS if a.keep else 0)
    for f in range(a.batch):
        ctx.upload_rgba8(f, rng.integers(0, 256, size=(a.h, a.w, 4), dtype=np.uint8))
    stream = torch.cuda.current_stream().cuda_stream or None
    ctx.upscale(stream)
    torch.cuda.synchronize()
    total = 0.0
    fpl = ctx.frames_per_launch
    for i, name in enumerate(ctx.stage_names()):
        ctx.launch_stage(i, stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ctx.launch_stage(i, stream)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.reps / fpl          # a launch covers the frames of one pass
        total += ms
        print(f"{name:32s} {ms:8.4f} ms/frame")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ctx.upscale(stream)
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / a.reps / a.batch
    print(f"{'sum of stages':32s} {total:8.4f} ms/frame")
    print(f"{'upscale()':32s} {ms:8.4f} ms/frame  = {1e3 / ms:8.1f} frames/s  (EXP={os.environ.get('SRX_EXP', '0')})")


if __name__ == "__main__":
    main()
