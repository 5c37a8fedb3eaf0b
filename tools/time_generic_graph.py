"""Times the sample network built from the generic layers (fp16, both layouts) on one 1080p frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import superresolution_amd as S
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import synthetic_rgba
W, H = 1920, 1080
img = synthetic_rgba(W, H, 3)
for flags, name in ((S.FLAG_GENERIC_LAYERS, "fp16 NCHW"), (S.FLAG_GENERIC_LAYERS | S.FLAG_NHWC, "fp16 NHWC")):
    with S.SuperResolutionContext(W, H, flags=flags) as ctx:
        ctx.upload_rgba8(0, img)
        ctx.upscale(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5): ctx.upscale()
        ctx.sync()
        print(name, "%.2f ms per 1080p frame" % ((time.perf_counter() - t0) / 5 * 1e3))
