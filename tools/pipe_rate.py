#!/usr/bin/env python3
"""Host-to-host rate of the frame pipe at 1080p from Python, with either HIP runtime (development helper).

  python tools/pipe_rate.py            # libsrx.so shares torch's bundled HIP runtime (as bench.py does)
  python tools/pipe_rate.py --system   # torch is kept out of the process: the system's /opt/rocm runtime
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--system" in sys.argv:
    sys.modules["torch"] = None          # makes `import torch` fail inside superresolution_amd._capi.lib()
import numpy as np                                                   # noqa: E402
import superresolution_amd as S                                      # noqa: E402

W, H, per_slot, slots, rounds = 1920, 1080, 8, 3, 40
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
with S.SuperResolutionContext(W, H, batch=per_slot * slots) as ctx, S.FramePipe(ctx, per_slot, 3, 3) as pipe:
    for s in range(slots):
        pipe.input(s)[:] = frame
    done = []
    jobs = rounds * slots
    for j in range(jobs + slots):
        s = j % slots
        if j >= slots:
            pipe.wait(s)
            done.append(time.perf_counter())
        if j < jobs:
            pipe.submit(s)
    n = (len(done) - 1 - 2 * slots) * per_slot
    print(f"{'system' if '--system' in sys.argv else 'torch'} runtime: {n / (done[-1] - done[2 * slots]):.1f} frames/s host to host")
