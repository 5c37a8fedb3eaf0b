#!/usr/bin/env python3
"""One resident image of the sample's own sizes and a few larger ones: GPU time of an upscale (both fused kernels back to
back, HIP events over 200 calls), wall time of a call + synchronize, and each stage alone.  On the GPU box:
  python tools/small_image_latency.py"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import superresolution_amd as S
rng = np.random.default_rng(1)
for (w, h) in [(91, 91), (128, 128), (256, 256), (512, 512), (640, 360), (1280, 720), (1920, 1080)]:
    f = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    with S.SuperResolutionContext(w, h) as ctx:
        ctx.upload_rgba8(0, f)
        st = torch.cuda.current_stream().cuda_stream or None
        for _ in range(20): ctx.upscale(st)
        torch.cuda.synchronize()
        # back-to-back launches timed with events: per-image GPU time without host launch gaps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ctx.upscale(st)
        e1.record(); e1.synchronize()
        gpu = e0.elapsed_time(e1) / 200
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); ctx.upscale(st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        stages = []
        for i, n in enumerate(ctx.stage_names()):
            e0.record()
            for _ in range(100): ctx.launch_stage(i, st)
            e1.record(); e1.synchronize()
            stages.append(round(e0.elapsed_time(e1) / 100 * 1e3, 1))
        print(f"{w}x{h}: back-to-back {gpu*1e3:.1f} us, call+sync median {sorted(ts)[50]*1e6:.1f} us, stages {stages} us, ideal at 1080p rate {w*h/(1920*1080)*620:.1f} us")
