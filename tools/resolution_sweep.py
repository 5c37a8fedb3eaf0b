import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import superresolution_amd as S
rng = np.random.default_rng(0)
for (w, h, batch) in [(1280, 720, 66), (1280, 720, 16), (1000, 1000, 32), (1920, 1080, 64), (640, 360, 128), (2560, 1440, 16), (3840, 2160, 8)]:
    with S.SuperResolutionContext(w, h, batch=batch) as ctx:
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        for f in range(batch):
            ctx.upload_rgba8(f, img)
        ctx.upscale(); ctx.sync()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.upscale()
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps / batch
        eq = dt * (1920 * 1080) / (w * h)
        print(f"{w}x{h} batch {batch}: frames per pass {ctx.frames_per_launch}, {dt * 1e3:.4f} ms/frame, {eq * 1e3:.3f} ms per 1080p-equivalent, {4 * w * h / dt / 1e9:.2f} output GP/s")
