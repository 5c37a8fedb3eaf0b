"""CPU tests of the shipped library's host side: the C ABI loads and exports every declared
symbol, and its weight path (parse, fold, truncate, MFMA packing) agrees with the oracle.
No compute entry point is called here (there is no GPU in this tier)."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import superresolution_amd as S
from superresolution_amd import _capi
from conftest import ASSETS, GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    L = S.lib()
    header = open(os.path.join(ROOT, "include", "srx.h")).read()
    declared = set(re.findall(r"\b(srx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_capi.SYMBOLS)
    for name in declared:
        assert getattr(L, name) is not None


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "lib_path", lambda: "/nonexistent/libsrx.so")
    with pytest.raises(S.SrxError):
        _capi.lib()


def test_fold_matches_oracle_and_fixture(oracle, oracle_weights):
    g = np.load(os.path.join(GOLDEN, "weights_fp16.npz"))
    for li, (conv, scale, shift, sizes) in enumerate(oracle.LAYERS):
        f, b = S.fold_weights(li)
        fo, bo = oracle_weights.create_weight_tensors(conv, scale, shift, sizes)
        assert np.array_equal(f, fo) and np.array_equal(f, g[f"filter{li}"])
        if bo is None:
            assert b is None
        else:
            assert np.array_equal(b, bo)


def test_weight_errors(tmp_path, capfd):
    L = S.lib()
    n = C.c_size_t()
    assert L.srx_fold_weights(str(tmp_path / "nope.bin").encode(), 0, None, 0, None, 0, C.byref(n), None) == 2
    assert "Unable to open weight file" in capfd.readouterr().err
    assert b"loadWeights" in L.srx_last_error()
    p = tmp_path / "neg.bin"
    p.write_bytes(struct.pack("<i", -5))
    assert L.srx_fold_weights(str(p).encode(), 0, None, 0, None, 0, C.byref(n), None) == 2
    assert "Invalid weight map file" in capfd.readouterr().err
    p = tmp_path / "long.bin"
    p.write_bytes(struct.pack("<iI", 1, 300) + b"y" * 400)
    assert L.srx_fold_weights(str(p).encode(), 0, None, 0, None, 0, C.byref(n), None) == 2
    assert "name_len exceeds c_bufferLength" in capfd.readouterr().err
    p = tmp_path / "trunc.bin"
    p.write_bytes(struct.pack("<iI", 1, 13) + b"conv1/weights" + struct.pack("<I", 2400) + b"\0" * 100)
    assert L.srx_fold_weights(str(p).encode(), 0, None, 0, None, 0, C.byref(n), None) == 2
    assert "Invalid tensor data" in capfd.readouterr().err
    # a well-formed file that lacks the layers
    p = tmp_path / "empty.bin"
    p.write_bytes(struct.pack("<i", 0))
    assert L.srx_fold_weights(str(p).encode(), 0, None, 0, None, 0, C.byref(n), None) == 2
    assert L.srx_fold_weights(S.default_weights_path().encode(), 9, None, 0, None, 0, C.byref(n), None) == 1
    small = np.zeros(4, np.uint16)
    assert L.srx_fold_weights(S.default_weights_path().encode(), 1, small.ctypes.data, 4, None, 0, None, None) == 1


def _unpack3x3(packed, cout, cin):
    """Invert the fragment layout back to OIHW."""
    tiles, cch = (cout + 31) // 32, cin // 16
    p = packed.reshape(tiles, 9, cch, 64, 8)
    out = np.zeros((tiles * 32, cin, 9), np.uint16)
    for lane in range(64):
        o, h = lane & 31, lane >> 5
        for j in range(8):
            # [tile, tap, cc] -> channel 16cc + 8h + j
            out[o::32, (8 * h + j)::16, :] = p[:, :, :, lane, j].transpose(0, 2, 1).reshape(tiles, cch, 9)
    return out


def test_packed_3x3_layout():
    for li, (cout, cin) in {1: (64, 32), 2: (64, 64), 4: (32, 32), 5: (32, 32), 6: (3, 32)}.items():
        f, _ = S.fold_weights(li)
        packed = S.packed_weights(li)
        tiles = (cout + 31) // 32
        assert packed.size == tiles * 9 * (cin // 16) * 64 * 8
        back = _unpack3x3(packed, cout, cin)
        assert np.array_equal(back[:cout], f.reshape(cout, cin, 9))
        assert not back[cout:].any()          # padded output rows are zero


def test_packed_conv1_layout():
    f, _ = S.fold_weights(0)
    w = f.reshape(32, 3, 25)
    p = S.packed_weights(0).reshape(7, 64, 8)
    for s in range(7):
        for lane in (0, 5, 31, 32, 40, 63):
            o, h = lane & 31, lane >> 5
            for j in range(8):
                tap, c = 4 * s + 2 * h + (j >> 2), j & 3
                want = w[o, c, tap] if (tap < 25 and c < 3) else 0
                assert p[s, lane, j] == want


def test_packed_up1_phase_filters(oracle):
    """Phase filters = sums of the 5x5 taps that read the same low-resolution pixel, rounded once."""
    f, _ = S.fold_weights(3)
    w = f.view(np.float16).astype(np.float64).reshape(32, 64, 5, 5)
    packed = S.packed_weights(3).reshape(4, -1)
    groups = {0: [[0, 1], [2, 3], [4]], 1: [[0], [1, 2], [3, 4]]}
    for py in range(2):
        for px in range(2):
            ph = np.zeros((32, 64, 3, 3))
            for a, rows in enumerate(groups[py]):
                for b, cols in enumerate(groups[px]):
                    ph[:, :, a, b] = w[:, :, rows][:, :, :, cols].sum(axis=(2, 3))
            want = ph.astype(np.float32).astype(np.float16).view(np.uint16).reshape(32, 64, 9)
            back = _unpack3x3(packed[py * 2 + px], 32, 64)
            assert np.array_equal(back, want)
    # and the direct/phase equivalence itself, on exact (unrounded) sums: a 5x5 conv of the
    # NN-upsampled signal equals the phase 3x3 conv of the low-resolution signal
    rng = np.random.default_rng(0)
    x = rng.standard_normal((6, 7))
    k = rng.standard_normal((5, 5))
    up = np.pad(np.repeat(np.repeat(x, 2, 0), 2, 1), 2)
    direct = np.array([[(up[Y:Y + 5, X:X + 5] * k).sum() for X in range(14)] for Y in range(12)])
    xp = np.pad(x, 1)
    for py in range(2):
        for px in range(2):
            ph = np.array([[k[np.ix_(r, c)].sum() for c in groups[px]] for r in groups[py]])
            got = np.array([[(xp[y:y + 3, xx:xx + 3] * ph).sum() for xx in range(7)] for y in range(6)])
            assert np.allclose(got, direct[py::2, px::2])


def test_fused_k32_layout():
    """conv2 / conv3 / conv4 / conv5 as v_mfma_f32_16x16x32_f16 A fragments: lane (r, q) = row r of its half,
    channels 8q..8q+7."""
    for li, (cout, cin) in {1: (64, 32), 2: (64, 64), 4: (32, 32), 5: (32, 32)}.items():
        f, _ = S.fold_weights(li)
        w = f.reshape(cout, cin, 9)
        kch = cin // 32
        p = S.fused_weights(li).reshape(cout // 32, 9 * kch, 2, 64, 8)
        for lane in range(64):
            r, q = lane & 15, lane >> 4
            for tile in range(cout // 32):
                for half in range(2):
                    # the rows of the two halves are interleaved, so that a lane group's four D rows of both halves
                    # are eight consecutive outputs
                    o = tile * 32 + 8 * (r >> 2) + 4 * half + (r & 3)
                    for kc in range(kch):
                        want = w[o, kc * 32 + 8 * q: kc * 32 + 8 * q + 8, :]           # [8, 9]
                        got = p[tile, kc::kch, half, lane, :]                             # [9 taps, 8]
                        assert np.array_equal(got, want.T)


def test_fused_up1_row_shared_layout():
    """conv_up1 as the fused kernel holds it: per wave (output half, column phase px) 18 S fragments (filter rows 0, 2,
    4) and 12 T fragments (filter rows 1, 3); the three column taps of a phase are the sums of the 5x5 taps that read
    the same input column (px = 0: 0+1, 2+3, 4; px = 1: 0, 1+2, 3+4), rounded once to fp16.  Folding the ROWS of the
    two output row phases from these (out(2y) = S(y) + T(y), out(2y+1) = S(y) + T(y+1)) gives the four 3x3 phase
    filters of the layer-wise kernel up to that rounding."""
    f, _ = S.fold_weights(3)
    w = f.view(np.float16).astype(np.float64).reshape(32, 64, 5, 5)
    cols = [np.stack([w[..., 0] + w[..., 1], w[..., 2] + w[..., 3], w[..., 4]], -1),
            np.stack([w[..., 0], w[..., 1] + w[..., 2], w[..., 3] + w[..., 4]], -1)]       # [px][32, 64, 5, 3]
    p = S.fused_weights(3).reshape(4, 30, 64, 8)
    for wave in range(4):
        px, oh = wave & 1, wave >> 1
        want = cols[px].astype(np.float16)
        for frag in range(30):
            g = frag if frag < 18 else frag - 18
            slot, j, kc = g // 6, (g % 6) // 2, g % 2
            i = 2 * slot if frag < 18 else 2 * slot + 1
            for lane in (0, 5, 15, 16, 38, 63):
                o, q = oh * 16 + (lane & 15), lane >> 4
                got = p[wave, frag, lane].view(np.float16)
                assert np.array_equal(got, want[o, kc * 32 + 8 * q: kc * 32 + 8 * q + 8, i, j]), (wave, frag, lane)
    # rows folded per output row phase: py = 0 reads input rows (y-1, y, y+1) with filter rows (0+1, 2+3, 4), py = 1 with (0, 1+2, 3+4)
    a = S.packed_weights(3).reshape(4, -1)
    for py in range(2):
        for px in range(2):
            c = cols[px].astype(np.float16).astype(np.float64)
            rows = ([c[:, :, 0] + c[:, :, 1], c[:, :, 2] + c[:, :, 3], c[:, :, 4]] if py == 0 else
                    [c[:, :, 0], c[:, :, 1] + c[:, :, 2], c[:, :, 3] + c[:, :, 4]])
            folded = np.stack(rows, 2).reshape(32, 64, 9)
            phase = _unpack3x3(a[py * 2 + px], 32, 64).view(np.float16).astype(np.float64)
            assert np.abs(folded - phase).max() <= 2 ** -8 * max(1.0, np.abs(phase).max())


def test_fused_conv6_rotating_layout():
    """conv6 streams input rows: for input row R (rho = R mod 4) group g of the 16 MFMA rows gets filter row
    2 (g = rho - 1), 1 (g = rho) or 0 (g = rho + 1) of output channel c = row & 3; everything else is zero."""
    f, _ = S.fold_weights(6)
    w = f.reshape(3, 32, 3, 3)
    p = S.fused_weights(6).reshape(4, 3, 64, 8)
    for rho in range(4):
        for j in range(3):
            for lane in range(64):
                row, q = lane & 15, lane >> 4
                g, c = row >> 2, row & 3
                i = {(rho + 3) & 3: 2, rho: 1, (rho + 1) & 3: 0}.get(g)
                want = w[c, 8 * q:8 * q + 8, i, j] if (c < 3 and i is not None) else np.zeros(8, np.uint16)
                assert np.array_equal(p[rho, j, lane], want)
    # summed over the three input rows that touch it, every output row gets each filter row exactly once
    for y in range(4):
        rows = sorted({(rho + 3) & 3: 2, rho: 1, (rho + 1) & 3: 0}[y] for rho in range(4) if y in ((rho + 3) & 3, rho, (rho + 1) & 3))
        assert rows == [0, 1, 2]


def test_fused_conv1_layout():
    """conv1 in the fused kernel: seven K-steps of four taps (row-major over the 5x5) x four channels; lane (o, h)
    holds taps 4s + 2h and 4s + 2h + 1; taps 25 .. 27 and the fourth channel are zero.  The same fragments as the
    layer-wise kernel's."""
    f, _ = S.fold_weights(0)
    w = f.reshape(32, 3, 25)
    p = S.fused_weights(0).reshape(7, 64, 8)
    assert np.array_equal(p.ravel(), S.packed_weights(0).ravel())
    for s in range(7):
        for lane in (0, 9, 31, 32, 50, 63):
            o, h = lane & 31, lane >> 5
            for jj in range(8):
                tap, c = 4 * s + 2 * h + (jj >> 2), jj & 3
                want = w[o, c, tap] if (tap < 25 and c < 3) else 0
                assert p[s, lane, jj] == want


def test_layer_factories_reproduce_the_reference_buffer_plan():
    """createResources (Content/SuperResolution.cpp:172-275) walks the layers with running maxima of the
    input / output byte sizes; at 1920x1080 fp16 that gives the two ping-pong buffers of 0.53 and 1.06 GB."""
    P = 1920 * 1080
    sizes, ib, ob = (1, 3, 1080, 1920), 0, 0
    out0, ib, ob = S.layer_plan(1, 0, sizes, (2, 2, 0, 0), ib, ob)
    assert out0 == (1, 3, 2160, 3840)
    cur = sizes
    filters = [(32, 3, 5, 5), (64, 32, 3, 3), (64, 64, 3, 3), (32, 64, 5, 5), (32, 32, 3, 3), (32, 32, 3, 3), (3, 32, 3, 3)]
    for li, f in enumerate(filters):
        if li == 3:
            cur, ib, ob = S.layer_plan(1, 0, cur, (2, 2, 0, 0), ib, ob)
            assert cur == (1, 64, 2160, 3840)
        cur, ib, ob = S.layer_plan(0, 0, cur, f, ib, ob)
        assert cur[1] == f[0] and cur[2:] == (2160, 3840) if li >= 3 else cur[2:] == (1080, 1920)
    assert S.layer_plan(2, 0, cur, (0, 0, 0, 0))[0] == (1, 3, 2160, 3840)
    assert ib == 64 * 4 * P * 2          # the upsampled conv3 tensor feeds conv_up1: 1.06 GB
    assert ob == 64 * 4 * P * 2
    # fp32 doubles them; a filter whose depth differs from the input channels, and zero scales, are rejected
    assert S.layer_plan(0, 1, (1, 3, 8, 8), (4, 3, 3, 3))[1:] == (3 * 64 * 4, 4 * 64 * 4)
    with pytest.raises(S.SrxError):
        S.layer_plan(0, 0, (1, 3, 8, 8), (4, 2, 3, 3))
    with pytest.raises(S.SrxError):
        S.layer_plan(1, 0, (1, 3, 8, 8), (0, 2, 0, 0))
    with pytest.raises(S.SrxError):
        S.layer_plan(7, 0, (1, 3, 8, 8), (1, 1, 1, 1))


def test_create_rejects_sizes_before_touching_a_device():
    """Size validation is host-only: 2^27 input pixels would make one output frame exactly 2^31 bytes, which
    the kernels' 32-bit byte offsets cannot address."""
    L = S.lib()
    h = C.c_void_p()
    wp = S.default_weights_path().encode()
    for w, hh, batch in [(16384, 8192, 1), (8192, 16384, 1), (16385, 16, 1), (16, 16385, 1), (0, 16, 1), (16, 0, 1), (16, 16, 0)]:
        assert L.srx_create(C.byref(h), 0, wp, w, hh, batch, 0) == 1, (w, hh, batch)
        assert not h.value
        assert b"fewer than 2^27 pixels" in L.srx_last_error()


def test_up1_phase_form_with_all_channels():
    """The four-phase form of conv_up1 with the library's own phase filters (unpacked from the fragments the
    kernels load), all 64 x 32 channels, borders included.  With exact (float64, unrounded) tap sums it IS the
    reference's direct 5x5 convolution of the nearest-neighbour-upsampled tensor; the fp16 rounding of the summed
    taps is the only difference of the product path, and its size is stated here."""
    f, b = S.fold_weights(3)
    w = f.view(np.float16).astype(np.float64).reshape(32, 64, 5, 5)
    groups = {0: [[0, 1], [2, 3], [4]], 1: [[0], [1, 2], [3, 4]]}
    rng = np.random.default_rng(3)
    # activations like conv3's output: non-negative, fp16, sparse, magnitudes of a few units
    x = np.maximum(rng.standard_normal((64, 9, 11)) * 1.5, 0).astype(np.float16).astype(np.float64)
    up = np.pad(np.repeat(np.repeat(x, 2, 1), 2, 2), ((0, 0), (2, 2), (2, 2)))
    direct = np.zeros((32, 18, 22))
    for i in range(5):
        for j in range(5):
            direct += np.einsum("oc,chw->ohw", w[:, :, i, j], up[:, i:i + 18, j:j + 22])
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    packed = S.packed_weights(3).reshape(4, -1)
    exact, rounded = np.zeros_like(direct), np.zeros_like(direct)
    for py in range(2):
        for px in range(2):
            lib = _unpack3x3(packed[py * 2 + px], 32, 64).view(np.float16).astype(np.float64).reshape(32, 64, 3, 3)
            for a, rows in enumerate(groups[py]):
                for c, cols in enumerate(groups[px]):
                    tap = w[:, :, rows][:, :, :, cols].sum(axis=(2, 3))
                    win = xp[:, a:a + 9, c:c + 11]
                    exact[:, py::2, px::2] += np.einsum("oc,chw->ohw", tap, win)
                    rounded[:, py::2, px::2] += np.einsum("oc,chw->ohw", lib[:, :, a, c], win)
    assert np.abs(exact - direct).max() < 1e-9
    err = np.abs(rounded - direct).max()
    # measured: 1.5e-3 for results of magnitude up to 5.6, i.e. under one fp16 ulp (3.9e-3) of such a result
    # and under a tenth of the 2e-2 the path is specified to
    assert err < 3e-3, err
    assert np.abs(direct).max() > 4


def test_diagnostic_build_exports_the_same_abi():
    """libsrx_diag.so (make DIAG=1) is the same library plus diagnostics: every declared symbol is there too."""
    import ctypes as C
    from superresolution_amd import _capi
    path = os.path.join(os.path.dirname(_capi.lib_path()), "libsrx_diag.so")
    if not os.path.exists(path):
        pytest.skip("the diagnostic library has not been built")
    L = C.CDLL(path)
    for name in _capi.SYMBOLS:
        assert hasattr(L, name), name
