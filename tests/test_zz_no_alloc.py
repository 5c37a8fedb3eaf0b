"""srx_upscale_static, the entry point that allocates nothing: its output against srx_upscale's on the same
image, byte for byte (the same two kernels run with the same plan; only the home of conv_up1's tensor differs:
a workspace inside the library's device image instead of memory the ctx allocated).  Two calls in a row, the
second on a smaller image, so the second call meets the first one's leftovers in the shared workspace, then a
third on a size whose border falls where the first call wrote.  torch only provides the caller's device images."""
import ctypes as C

import numpy as np
import pytest

import superresolution_amd as S
from conftest import synthetic_rgba

pytestmark = pytest.mark.gpu


def _reference(rgba):
    h, w = rgba.shape[:2]
    with S.SuperResolutionContext(w, h) as ctx:
        return ctx.run(rgba)


def test_static_workspace_entry_point_matches_srx_upscale():
    import torch
    sizes = [(200, 120), (131, 77), (160, 120), (1, 1), (61, 33)]
    frames = [synthetic_rgba(w, h, 7700 + i) for i, (w, h) in enumerate(sizes)]
    want = [_reference(f) for f in frames]
    with S.SuperResolutionContext(16, 16) as ctx:           # the ctx only lends its weights
        for f, ref in zip(frames, want):
            h, w = f.shape[:2]
            src = torch.from_numpy(f).cuda()
            dst = torch.full((2 * h, 2 * w, 4), 7, dtype=torch.uint8, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream or None
            ctx.upscale_static(src.data_ptr(), w * 4, dst.data_ptr(), 2 * w * 4, w, h, stream)
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), ref), (w, h)
        # pitched caller images: rows further apart than the pixels need, the gap must stay untouched
        f, ref = frames[1], want[1]
        h, w = f.shape[:2]
        src = torch.zeros((h, w + 5, 4), dtype=torch.uint8, device="cuda")
        src[:, :w] = torch.from_numpy(f).cuda()
        dst = torch.full((2 * h, 2 * w + 3, 4), 9, dtype=torch.uint8, device="cuda")
        ctx.upscale_static(src.data_ptr(), (w + 5) * 4, dst.data_ptr(), (2 * w + 3) * 4, w, h, None)
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        assert np.array_equal(got[:, :2 * w], ref) and (got[:, 2 * w:] == 9).all()


def test_static_workspace_entry_point_rejects_what_it_cannot_hold():
    import torch
    L = S.lib()
    assert L.srx_static_scratch_bytes() == 64 << 20
    with S.SuperResolutionContext(16, 16) as ctx:
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
        p = buf.data_ptr()
        assert L.srx_upscale_static(ctx._h, p, 1280 * 4, p, 2560 * 4, 1280, 720, None) == 1      # SRX_ERR_INVALID
        assert b"too large for the static workspace" in L.srx_last_error()
        assert L.srx_upscale_static(ctx._h, None, 64, p, 128, 16, 16, None) == 1
        assert L.srx_upscale_static(ctx._h, p, 60, p, 128, 16, 16, None) == 1                     # pitch below a row
        assert L.srx_upscale_static(ctx._h, p + 1, 64, p, 128, 16, 16, None) == 1                 # misaligned
    with S.SuperResolutionContext(16, 16, flags=S.FLAG_GENERIC_LAYERS) as ctx:
        assert L.srx_upscale_static(ctx._h, p, 64, p, 128, 16, 16, None) == 5                     # SRX_ERR_STATE


def test_static_workspace_largest_size():
    """640x360 is the documented largest common size: it must fit and agree."""
    import torch
    w, h = 640, 360
    f = synthetic_rgba(w, h, 7800)
    ref = _reference(f)
    with S.SuperResolutionContext(32, 32) as ctx:
        src = torch.from_numpy(f).cuda()
        dst = torch.zeros((2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            ctx.upscale_static(src.data_ptr(), w * 4, dst.data_ptr(), 2 * w * 4, w, h, None)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), ref)


def test_static_entry_point_under_hip_graph_capture():
    """What the entry point is for: under stream capture nothing can be allocated.  The call is captured into a
    HIP graph (through torch's wrapper of hipGraph) and replayed twice, the second time on new input bytes; both
    replays must give the bytes srx_upscale gives.  (A graph whose workspace was cleared by a memset node instead
    of a kernel node fails the second replay: the memset overlaps the first kernel.)"""
    import torch
    w, h = 150, 101
    frames = [synthetic_rgba(w, h, 7900 + i) for i in range(2)]
    want = [_reference(f) for f in frames]
    with S.SuperResolutionContext(16, 16) as ctx:
        src = torch.from_numpy(frames[0]).cuda()
        dst = torch.zeros((2 * h, 2 * w, 4), dtype=torch.uint8, device="cuda")
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ctx.upscale_static(src.data_ptr(), w * 4, dst.data_ptr(), 2 * w * 4, w, h, side.cuda_stream)
        for f, ref in zip(frames, want):
            src.copy_(torch.from_numpy(f).cuda())
            dst.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(dst.cpu().numpy(), ref)
