#!/usr/bin/env python3
"""A/B timing of several builds of libsrx.so in ONE process, rounds interleaved (development helper).

  python tools/ab_libs.py base=gpurun_ab/libsrx_base.so new=superresolution_amd/libsrx.so [--rounds 7 --reps 10]
  python tools/ab_libs.py narrow=gpurun_ab/a.so,SRX_HR_WIDE=0 wide=gpurun_ab/b.so,SRX_HR_WIDE=1

A build may be followed by KEY=VALUE pairs: they are put into the environment before that build is loaded and its
context created, which is when a diagnostic library reads its switches (once).  Two variants of ONE diagnostic
library need two copies of the file under different names.

Per build and round: the duration of one launch of each stage and of a whole upscale of a pass of 1080p
frames (HIP events on the launch stream).  Prints median and minimum per build, and checks that the builds
produce the same image bytes (or reports how far apart they are)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+", help="name=path")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    import torch
    from superresolution_amd import _capi
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, size=(a.h, a.w, 4), dtype=np.uint8) for _ in range(min(a.batch, 2))]
    stream = torch.cuda.current_stream().cuda_stream or None
    builds = []
    for spec in a.libs:
        name, path = spec.split("=", 1)
        path, *env = path.split(",")
        for kv in env:
            k, v = kv.split("=", 1)
            os.environ[k] = v
        L = _capi.load_library(os.path.abspath(path))
        ctx = _capi.SuperResolutionContext(a.w, a.h, batch=a.batch, library=L)
        for f in range(a.batch):
            ctx.upload_rgba8(f, frames[f % len(frames)])
        ctx.upscale(stream)
        torch.cuda.synchronize()
        builds.append((name, ctx, {}))
    ref = builds[0][1].tensor_to_image(0)
    for name, ctx, _ in builds[1:]:
        img = ctx.tensor_to_image(0)
        d = np.abs(img.astype(int) - ref.astype(int))
        print(f"{name} vs {builds[0][0]}: max |diff| {d.max()}, identical {float((d == 0).mean()):.5f}")

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / a.reps

    for _ in range(a.rounds):
        for name, ctx, res in builds:
            for i, st in enumerate(ctx.stage_names()):
                res.setdefault(st, []).append(timed(lambda: ctx.launch_stage(i, stream)))
            res.setdefault("upscale", []).append(timed(lambda: ctx.upscale(stream)))
    for name, ctx, res in builds:
        print(f"== {name}: frames per launch {ctx.frames_per_launch}")
        for k, v in res.items():
            v = sorted(v)
            print(f"   {k:34s} median {v[len(v) // 2]:8.4f} ms   min {v[0]:8.4f} ms")
        ctx.close()


if __name__ == "__main__":
    main()
