"""GPU tests of the generic layers (ML::Util::Create*Layer operators on dense tensors) and of the
graph built from them: fp32 tensor mode, the NHWC switch, and the literal fp16 reference graph.

The oracle's fp32 mode (oracle/sr_oracle.c, dtype 1) is the checker for the graph; single layers
with shapes the sample network never uses are checked against a numpy restatement written here.
DirectML itself cannot be run in this environment: these results are parity unpinned against it.
"""
import numpy as np
import pytest
import torch

import superresolution_amd as S
from conftest import synthetic_rgba

pytestmark = pytest.mark.gpu

FP32_TOL = 2e-4          # fp32 accumulation order differs between the kernel and the oracle
FP16_TOL = 2e-2


def image_close(a, b):
    d = np.abs(a[..., :3].astype(int) - b[..., :3].astype(int))
    return (d <= 1).mean() >= 0.999 and (a[..., 3] == 255).all()


@pytest.fixture(scope="module")
def fp32_model(oracle, oracle_weights):
    return oracle.Model(oracle_weights, 1, 0)


@pytest.fixture(scope="module")
def direct_model(oracle, oracle_weights):
    return oracle.Model(oracle_weights, 0, 0)


@pytest.mark.parametrize("flags", [S.FLAG_FP32, S.FLAG_FP32 | S.FLAG_NHWC])
@pytest.mark.parametrize("w,h", [(40, 27), (33, 21)])
def test_fp32_graph_against_oracle(fp32_model, flags, w, h):
    rgba = synthetic_rgba(w, h, 77 + w)
    ref = fp32_model.run(rgba, keep_tensors=True)
    with S.SuperResolutionContext(w, h, flags=flags) as ctx:
        out = ctx.run(rgba)
        for tid in range(9):
            got = ctx.read_tensor(tid)
            assert got.dtype == np.float32
            assert np.abs(got - fp32_model.tensor(tid)).max() <= FP32_TOL, S.TENSOR_NAMES[tid]
    assert image_close(out, ref)


@pytest.mark.parametrize("flags", [S.FLAG_GENERIC_LAYERS, S.FLAG_GENERIC_LAYERS | S.FLAG_NHWC])
def test_fp16_generic_graph_against_oracle_direct_form(direct_model, flags):
    """The reference graph taken literally (up1 materialised, 5x5 conv_up1) in fp16 storage."""
    w, h = 37, 29
    rgba = synthetic_rgba(w, h, 5)
    ref = direct_model.run(rgba, keep_tensors=True)
    with S.SuperResolutionContext(w, h, flags=flags) as ctx:
        out = ctx.run(rgba)
        assert np.array_equal(ctx.read_tensor(0).astype(np.float32), direct_model.tensor(0))   # the pack step is exact
        for tid in range(1, 9):
            assert np.abs(ctx.read_tensor(tid).astype(np.float32) - direct_model.tensor(tid)).max() <= FP16_TOL
    assert image_close(out, ref)
    with S.SuperResolutionContext(w, h) as fast:          # and the fused MFMA path agrees with it
        assert image_close(fast.run(rgba), out)


def test_generic_graph_batch_frames_are_independent():
    frames = [synthetic_rgba(20, 14, 900 + i) for i in range(3)]
    with S.SuperResolutionContext(20, 14, flags=S.FLAG_FP32) as one:
        singles = [one.run(f) for f in frames]
    with S.SuperResolutionContext(20, 14, batch=3, flags=S.FLAG_FP32) as ctx:
        for i, f in enumerate(frames):
            ctx.upload_rgba8(i, f)
        ctx.upscale()
        for i in range(3):
            assert np.array_equal(ctx.tensor_to_image(i), singles[i])
        with pytest.raises(S.SrxError):
            ctx.upscale_frames(0, 1)


def conv_ref(x, w, b, relu):
    """numpy restatement: cross-correlation, zero padding ceil((k-1)/2) before, floor after (NCHW, float64)."""
    n, c, H, W = x.shape
    o, _, kh, kw = w.shape
    pt, pl = kh // 2, kw // 2
    xp = np.zeros((n, c, H + kh - 1, W + kw - 1))
    xp[:, :, pt:pt + H, pl:pl + W] = x
    y = np.zeros((n, o, H, W))
    for i in range(kh):
        for j in range(kw):
            y += np.einsum("nchw,oc->nohw", xp[:, :, i:i + H, j:j + W], w[:, :, i, j])
    if b is not None:
        y = np.maximum(y + b[None, :, None, None], 0.0)
    return y


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("shape", [(2, 5, 19, 45, 3, 4, 7, True), (1, 3, 9, 33, 17, 1, 1, False), (1, 20, 8, 8, 33, 7, 2, True),
                                   (1, 24, 12, 40, 40, 3, 3, True), (2, 64, 9, 70, 64, 5, 5, True), (1, 8, 17, 31, 1, 2, 2, False)])
def test_conv_layer_any_shape(dtype, layout, shape):
    n, c, H, W, o, kh, kw, bias = shape
    rng = np.random.default_rng(n * 100 + c)
    x = rng.uniform(-1, 1, (n, c, H, W)).astype(np.float16).astype(np.float64)
    w = (rng.uniform(-1, 1, (o, c, kh, kw)) / np.sqrt(c * kh * kw)).astype(np.float16).astype(np.float64)
    b = rng.uniform(-0.5, 0.5, o).astype(np.float16).astype(np.float64) if bias else None
    ref = conv_ref(x, w, b, bias)
    td = torch.float16 if dtype == 0 else torch.float32
    perm = (0, 2, 3, 1) if layout else (0, 1, 2, 3)
    dx = torch.tensor(x.transpose(perm).copy(), dtype=td, device="cuda")
    dw = torch.tensor(w.transpose(perm).copy(), dtype=td, device="cuda")       # O,kh,kw,I for NHWC
    db = torch.tensor(b, dtype=td, device="cuda") if bias else None
    dy = torch.empty((n, H, W, o) if layout else (n, o, H, W), dtype=td, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream or None
    S.layer_conv2d(dtype, layout, dx.data_ptr(), (n, c, H, W), dw.data_ptr(), (o, c, kh, kw),
                   db.data_ptr() if bias else None, dy.data_ptr(), stream)
    torch.cuda.synchronize()
    got = dy.cpu().double().numpy()
    got = got.transpose(0, 3, 1, 2) if layout else got
    assert np.abs(got - ref).max() <= (4e-3 if dtype == 0 else 2e-5)


@pytest.mark.parametrize("layout", [0, 1])
def test_upsample_and_add_layers(layout):
    rng = np.random.default_rng(3)
    n, c, H, W, sx, sy = 2, 3, 5, 7, 3, 2
    x = rng.uniform(-1, 1, (n, c, H, W)).astype(np.float32)
    perm = (0, 2, 3, 1) if layout else (0, 1, 2, 3)
    back = (0, 3, 1, 2) if layout else (0, 1, 2, 3)
    dx = torch.tensor(x.transpose(perm).copy(), device="cuda")
    shape = (n, H * sy, W * sx, c) if layout else (n, c, H * sy, W * sx)
    near, lin = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
    S.layer_upsample2d(1, layout, dx.data_ptr(), (n, c, H, W), sx, sy, 0, near.data_ptr())
    S.layer_upsample2d(1, layout, dx.data_ptr(), (n, c, H, W), sx, sy, 1, lin.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(near.cpu().numpy().transpose(back), x.repeat(sy, axis=2).repeat(sx, axis=3))
    # linear: half-pixel centres, clamped at the edges
    fy = np.clip((np.arange(H * sy) + 0.5) / sy - 0.5, 0, H - 1)
    fx = np.clip((np.arange(W * sx) + 0.5) / sx - 0.5, 0, W - 1)
    y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    y1, x1 = np.minimum(y0 + 1, H - 1), np.minimum(x0 + 1, W - 1)
    wy, wx = (fy - y0)[None, None, :, None], (fx - x0)[None, None, None, :]
    xd = x.astype(np.float64)
    ref = ((xd[:, :, y0][:, :, :, x0] * (1 - wx) + xd[:, :, y0][:, :, :, x1] * wx) * (1 - wy) +
           (xd[:, :, y1][:, :, :, x0] * (1 - wx) + xd[:, :, y1][:, :, :, x1] * wx) * wy)
    assert np.abs(lin.cpu().numpy().transpose(back) - ref).max() <= 1e-5
    a = torch.tensor(rng.uniform(-1, 1, 1000).astype(np.float16), device="cuda")
    b = torch.tensor(rng.uniform(-1, 1, 1000).astype(np.float16), device="cuda")
    expect = (a.float() + b.float()).half()
    S.layer_add(0, a.data_ptr(), b.data_ptr(), a.data_ptr(), 1000)           # in place, as the reference binds it
    torch.cuda.synchronize()
    assert torch.equal(a, expect)


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
def test_linear_upsample_against_torch_fixture(dtype, layout):
    """LINEAR mode in both storage types against tests/golden/upsample_linear.npz (torch-CPU float64 bilinear,
    align_corners=False, made by tests/golden/make_golden.py).  fp16: the kernel interpolates in fp32 and rounds
    once, so the result is within half an fp16 ulp of the float64 value (|x| <= 1: 2^-11) plus fp32 noise."""
    import os
    from conftest import GOLDEN
    fx = np.load(os.path.join(GOLDEN, "upsample_linear.npz"))
    td = torch.float16 if dtype == 0 else torch.float32
    perm = (0, 2, 3, 1) if layout else (0, 1, 2, 3)
    back = (0, 3, 1, 2) if layout else (0, 1, 2, 3)
    k = 0
    while f"x{k}" in fx:
        x, (sx, sy), want = fx[f"x{k}"], fx[f"scale{k}"], fx[f"y{k}"]
        n, c, H, W = x.shape
        dx = torch.tensor(x.transpose(perm).copy(), dtype=td, device="cuda")
        shape = (n, H * sy, W * sx, c) if layout else (n, c, H * sy, W * sx)
        out = torch.full(shape, float("nan"), dtype=td, device="cuda")
        S.layer_upsample2d(dtype, layout, dx.data_ptr(), (n, c, H, W), int(sx), int(sy), 1, out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().double().numpy().transpose(back)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= (2 ** -11 + 1e-5 if dtype == 0 else 1e-5), (k, dtype, layout)
        k += 1
    assert k == 5


def test_layer_argument_errors():
    x = torch.zeros(64, device="cuda")
    with pytest.raises(S.SrxError):      # filter depth != input channels
        S.layer_conv2d(1, 0, x.data_ptr(), (1, 4, 4, 4), x.data_ptr(), (2, 3, 3, 3), None, x.data_ptr())
    with pytest.raises(S.SrxError):      # filter side > 7
        S.layer_conv2d(1, 0, x.data_ptr(), (1, 1, 4, 4), x.data_ptr(), (1, 1, 9, 9), None, x.data_ptr())
    with pytest.raises(S.SrxError):      # unknown data type
        S.layer_upsample2d(2, 0, x.data_ptr(), (1, 1, 4, 4), 2, 2, 0, x.data_ptr())
    with pytest.raises(S.SrxError):
        S.layer_upsample2d(1, 0, x.data_ptr(), (1, 1, 4, 4), 0, 2, 0, x.data_ptr())
    with pytest.raises(S.SrxError):
        S.layer_add(1, None, x.data_ptr(), x.data_ptr(), 4)
