"""CPU tests of the host program's image codec (PNG in and out; BMP, TGA, binary PNM in) (superresolution_amd/csrc/host/ImageIO.cpp) against
PIL and, where oracle/_ref exists, against the reference's own stb_image compiled in place.
The contract is the decoded pixels and the channel rule of XUSG/Advanced/XUSGTextureLoader.h:17-32
(3-channel files load as RGBA, other channel counts load as they are)."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

from conftest import ASSETS, ROOT

LIB = os.path.join(ROOT, "superresolution_amd", "libsrximage.so")


@pytest.fixture(scope="module")
def img():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "superresolution_amd", "csrc", "host"), "../../libsrximage.so"])
    L = C.CDLL(LIB)
    L.srx_image_last_error.restype = C.c_char_p
    L.srx_image_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
    L.srx_image_load.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.srx_image_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L


def load(L, path):
    w, h, ch, req = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.srx_image_info(str(path).encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(req)) == 0, L.srx_image_last_error()
    out = np.empty((h.value, w.value, req.value), np.uint8)
    assert L.srx_image_load(str(path).encode(), out.ctypes.data, out.nbytes) == 0, L.srx_image_last_error()
    return out, ch.value, req.value


@pytest.mark.parametrize("name,channels", [("Sashimi128.png", 3), ("Sashimi.png", 3), ("xiaoyao.png", 4)])
def test_bundled_images(img, oracle, name, channels):
    path = os.path.join(ASSETS, name)
    got, ch, req = load(img, path)
    assert ch == channels and req == 4
    assert np.array_equal(got, np.asarray(Image.open(path).convert("RGBA")))
    R = oracle.ref()
    if R is not None:     # the reference's stb_image v2.30, compiled from where it lies
        w, h, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        p = R.ref_load_image(path.encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(r))
        ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, r.value)).copy()
        R.ref_free(p)
        assert c.value == ch and r.value == req and np.array_equal(ref, got)


@pytest.mark.parametrize("mode,kw", [("L", {}), ("LA", {}), ("RGB", {}), ("RGBA", {}), ("P", {}), ("RGB", {"interlace": True}),
                                     ("I;16", {}), ("1", {})])
def test_png_flavours(img, oracle, tmp_path, mode, kw):
    rng = np.random.default_rng(7)
    h, w = 37, 53
    if mode == "I;16":
        im = Image.fromarray(rng.integers(0, 65536, size=(h, w), dtype=np.uint16))
    elif mode == "1":
        im = Image.fromarray((rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8)).convert("1")
    elif mode == "P":
        im = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).quantize(17)
    else:
        n = {"L": 1, "LA": 2, "RGB": 3, "RGBA": 4}[mode]
        a = rng.integers(0, 256, size=(h, w, n), dtype=np.uint8)
        im = Image.fromarray(a[..., 0] if n == 1 else a, mode)
    path = tmp_path / f"t_{mode.replace(';', '')}.png"
    if kw.get("interlace"):
        pytest.importorskip("PIL")
        # PIL cannot write Adam7; build one by hand from its scanlines
        path = write_adam7(tmp_path / "adam7.png", np.asarray(im))
    else:
        im.save(path)
    got, ch, req = load(img, path)
    R = oracle.ref()
    if R is not None:
        wv, hv, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        p = R.ref_load_image(str(path).encode(), C.byref(wv), C.byref(hv), C.byref(c), C.byref(r))
        assert p
        ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(hv.value, wv.value, r.value)).copy()
        R.ref_free(p)
        assert (c.value, r.value) == (ch, req)
        assert np.array_equal(ref, got)
    else:
        want = {1: "L", 2: "LA", 4: "RGBA"}[req]
        ref = np.asarray(Image.open(path).convert(want)).reshape(h, w, req)
        if mode != "I;16":
            assert np.array_equal(ref, got)


def write_adam7(path, rgb):
    import struct
    import zlib
    h, w, _ = rgb.shape
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    raw = b""
    for x0, y0, dx, dy in passes:
        sub = rgb[y0::dy, x0::dx]
        if sub.size:
            for row in sub:
                raw += b"\0" + row.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    return path


@pytest.mark.parametrize("comp", [1, 2, 3, 4])
def test_write_png_roundtrip(img, tmp_path, comp):
    rng = np.random.default_rng(comp)
    h, w = 41, 29
    a = rng.integers(0, 256, size=(h, w, comp), dtype=np.uint8)
    a[5:20, 3:25] = 77        # flat area: exercises the filter choice
    path = tmp_path / f"w{comp}.png"
    assert img.srx_image_write_png(str(path).encode(), w, h, comp, a.ctypes.data, 0) == 0
    back = np.asarray(Image.open(path)).reshape(h, w, comp)
    assert np.array_equal(back, a)
    # with a row stride (a sub-rectangle of a wider image)
    wide = np.zeros((h, w + 11, comp), np.uint8)
    wide[:, :w] = a
    assert img.srx_image_write_png(str(path).encode(), w, h, comp, wide.ctypes.data, (w + 11) * comp) == 0
    assert np.array_equal(np.asarray(Image.open(path)).reshape(h, w, comp), a)


def test_bad_files(img, tmp_path):
    w, h, ch, req = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert img.srx_image_info(str(tmp_path / "none.png").encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(req)) == 1
    p = tmp_path / "junk.png"
    p.write_bytes(b"not a png at all, just bytes" * 4)
    assert img.srx_image_info(str(p).encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(req)) == 1
    good = open(os.path.join(ASSETS, "xiaoyao.png"), "rb").read()
    p = tmp_path / "cut.png"
    p.write_bytes(good[:3000])       # inside the single IDAT chunk
    out = np.empty(91 * 91 * 4, np.uint8)
    assert img.srx_image_load(str(p).encode(), out.ctypes.data, out.nbytes) == 1


def ref_load(R, path):
    w, h, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    p = R.ref_load_image(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(r))
    assert p
    ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, r.value)).copy()
    R.ref_free(p)
    return ref, c.value, r.value


@pytest.mark.parametrize("ext,mode,kw", [("bmp", "RGB", {}), ("bmp", "L", {}), ("bmp", "P", {}), ("bmp", "RGBA", {}),
                                         ("tga", "RGB", {}), ("tga", "RGBA", {}), ("tga", "L", {}),
                                         ("tga", "RGB", {"compression": "tga_rle"}), ("tga", "RGBA", {"compression": "tga_rle"}),
                                         ("tga", "L", {"compression": "tga_rle"}), ("ppm", "RGB", {}), ("pgm", "L", {})])
def test_other_formats_the_reference_loader_accepts(img, oracle, tmp_path, ext, mode, kw):
    """BMP, TGA (plain and run-length) and binary PNM as PIL writes them: the decoded pixels and the
    channel rule must equal the reference's stb_image (where it could be compiled) and PIL."""
    rng = np.random.default_rng(11)
    h, w = 29, 43                      # odd width: BMP rows are padded to four bytes
    n = {"L": 1, "P": 3, "RGB": 3, "RGBA": 4}[mode]
    a = rng.integers(0, 256, size=(h, w, n), dtype=np.uint8)
    a[:7, :9] = a[0, 0]                # a flat patch, so that the run-length files have runs
    im = Image.fromarray(a[..., 0] if n == 1 else a, "RGB" if mode == "P" else mode)
    if mode == "P":
        im = im.quantize(23)
    path = tmp_path / f"t_{mode}_{len(kw)}.{ext}"
    im.save(path, **kw)
    got, ch, req = load(img, path)
    assert req == (4 if ch == 3 else ch)
    want_mode = {1: "L", 2: "LA", 4: "RGBA"}[req]
    pil = np.asarray(Image.open(path).convert(want_mode)).reshape(h, w, req)
    if (ext, mode) == ("bmp", "RGBA"):        # PIL drops the alpha of a 32-bit BMP when it reads it back; stb keeps it
        assert ch == 4 and np.array_equal(got[..., :3], pil[..., :3]) and np.array_equal(got[..., 3], a[..., 3])
    else:
        assert np.array_equal(got, pil)
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)


def test_pnm_with_comments(img, oracle, tmp_path):
    rng = np.random.default_rng(12)
    a = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    path = tmp_path / "c.ppm"
    path.write_bytes(b"P6\n# a comment\n7 5\n# another\n255\n" + a.tobytes())
    got, ch, req = load(img, path)
    assert (ch, req) == (3, 4) and np.array_equal(got[..., :3], a) and (got[..., 3] == 255).all()
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)
    # 16-bit samples: the reference's loader returns their low byte (a quirk of its 8-bit entry point on a
    # little-endian host); refused here rather than copied or silently corrected
    deep = tmp_path / "deep.ppm"
    deep.write_bytes(b"P6 7 5 65535\n" + bytes(7 * 5 * 6))
    w, h, c2, r2 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert img.srx_image_info(str(deep).encode(), C.byref(w), C.byref(h), C.byref(c2), C.byref(r2)) != 0
    assert b"16-bit" in img.srx_image_last_error()


def test_unsupported_flavours_are_refused(img, tmp_path):
    import struct
    # a BMP with bit fields (compression 3) and a 16-bit colour TGA: refused with a reason, not guessed at
    bmp = b"BM" + struct.pack("<IHHI", 70, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 2, 2, 1, 16, 3, 16, 0, 0, 0, 0) + bytes(16)
    (tmp_path / "bf.bmp").write_bytes(bmp)
    tga = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 2, 0, 16, 0]) + bytes(8)
    (tmp_path / "hi.tga").write_bytes(tga)
    for name in ("bf.bmp", "hi.tga"):
        w, h, ch, req = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        assert img.srx_image_info(str(tmp_path / name).encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(req)) != 0
        assert b"not supported" in img.srx_image_last_error()


@pytest.mark.parametrize("ext,mode,kw", [("tga", "P", {}), ("tga", "P", {"compression": "tga_rle"}), ("tga", "LA", {}), ("bmp", "1", {})])
def test_palette_and_grey_alpha_flavours(img, oracle, tmp_path, ext, mode, kw):
    """Colour-mapped TGA, grey + alpha TGA and 1-bit BMP as PIL writes them, against the reference's stb_image."""
    rng = np.random.default_rng(13)
    h, w = 19, 37
    if mode == "P":
        im = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB").quantize(29)
    elif mode == "LA":
        im = Image.fromarray(rng.integers(0, 256, size=(h, w, 2), dtype=np.uint8), "LA")
    else:
        im = Image.fromarray((rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8)).convert("1")
    path = tmp_path / f"t.{ext}"
    im.save(path, **kw)
    got, ch, req = load(img, path)
    want = {1: "L", 2: "LA", 4: "RGBA"}[req]
    assert np.array_equal(got, np.asarray(Image.open(path).convert(want)).reshape(h, w, req))
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)


def test_four_bit_bmp(img, oracle, tmp_path):
    import struct
    rng = np.random.default_rng(14)
    w, h = 13, 5
    idx = rng.integers(0, 16, size=(h, w), dtype=np.uint8)
    pal = rng.integers(0, 256, size=(16, 3), dtype=np.uint8)
    rows = b""
    for y in range(h - 1, -1, -1):                  # bottom-up
        packed = bytearray((w + 1) // 2)
        for x in range(w):
            packed[x >> 1] |= int(idx[y, x]) << (0 if x & 1 else 4)
        rows += bytes(packed) + bytes((-len(packed)) % 4)
    table = b"".join(bytes([int(b), int(g), int(r), 0]) for r, g, b in pal)
    off = 14 + 40 + len(table)
    data = b"BM" + struct.pack("<IHHI", off + len(rows), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 4, 0, len(rows), 0, 0, 16, 0) + table + rows
    path = tmp_path / "n.bmp"
    path.write_bytes(data)
    got, ch, req = load(img, path)
    assert (ch, req) == (3, 4) and np.array_equal(got[..., :3], pal[idx]) and (got[..., 3] == 255).all()
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)


@pytest.mark.parametrize("mode,size,kw", [("RGB", (43, 29), {"subsampling": 0}), ("RGB", (43, 29), {"subsampling": 1}),
                                          ("RGB", (43, 29), {"subsampling": 2}), ("RGB", (64, 48), {"subsampling": 2, "quality": 95}),
                                          ("RGB", (1, 1), {"subsampling": 2}), ("RGB", (17, 2), {"subsampling": 2, "quality": 30}),
                                          ("L", (43, 29), {}), ("L", (8, 8), {"quality": 100}),
                                          ("RGB", (131, 77), {"subsampling": 2, "optimize": True}),
                                          ("RGB", (131, 77), {"subsampling": 1, "restart_marker_blocks": 3})])
def test_baseline_jpeg_against_the_reference_decoder(img, oracle, tmp_path, mode, size, kw):
    """Baseline JPEG as PIL (libjpeg) writes it, every chroma subsampling, odd sizes, optimised
    Huffman tables, restart markers: the decoded bytes must equal the reference's stb_image, whose
    inverse DCT, chroma upsampling and colour conversion this decoder restates.  Without the
    compiled reference decoder only the plausibility of the result is checked."""
    rng = np.random.default_rng(21)
    w, h = size
    n = 3 if mode == "RGB" else 1
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(xx * 5 + yy * 3 + 40 * k) % 256 for k in range(n)], axis=2)
    a = ((smooth + rng.integers(0, 64, size=(h, w, n))) // 2 + 60).clip(0, 255).astype(np.uint8)
    im = Image.fromarray(a[..., 0] if n == 1 else a, mode)
    path = tmp_path / "t.jpg"
    try:
        im.save(path, format="JPEG", **kw)
    except TypeError:
        pytest.skip("this PIL cannot write that flavour")
    got, ch, req = load(img, path)
    assert (ch, req) == ((3, 4) if n == 3 else (1, 1)) and got.shape == (h, w, req)
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)
    else:
        d = got[..., :n].astype(float) - a
        assert 10 * np.log10(255.0 ** 2 / max((d ** 2).mean(), 1e-9)) > 22.0


@pytest.mark.parametrize("mode,size,kw", [("RGB", (43, 29), {"subsampling": 0}), ("RGB", (131, 77), {"subsampling": 2}),
                                          ("RGB", (64, 64), {"subsampling": 1, "quality": 95}), ("L", (57, 33), {}),
                                          ("RGB", (1, 1), {}), ("RGB", (200, 120), {"subsampling": 2, "quality": 20}),
                                          ("RGB", (131, 77), {"subsampling": 2, "restart_marker_blocks": 2})])
def test_progressive_jpeg_against_the_reference_decoder(img, oracle, tmp_path, mode, size, kw):
    """Progressive files (spectral selection and successive approximation, as libjpeg's default scan
    script writes them): byte-identical to the reference's stb_image."""
    rng = np.random.default_rng(22)
    w, h = size
    n = 3 if mode == "RGB" else 1
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(xx * 7 + yy * 2 + 50 * k) % 256 for k in range(n)], axis=2)
    a = ((smooth + rng.integers(0, 96, size=(h, w, n))) // 2 + 40).clip(0, 255).astype(np.uint8)
    im = Image.fromarray(a[..., 0] if n == 1 else a, mode)
    path = tmp_path / "p.jpg"
    im.save(path, format="JPEG", progressive=True, **kw)
    assert b"\xff\xc2" in path.read_bytes()
    got, ch, req = load(img, path)
    assert (ch, req) == ((3, 4) if n == 3 else (1, 1)) and got.shape == (h, w, req)
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)
    else:
        d = got[..., :n].astype(float) - a
        assert 10 * np.log10(255.0 ** 2 / max((d ** 2).mean(), 1e-9)) > 20.0


def test_unsupported_jpeg_flavours_are_refused(img, tmp_path):
    im = Image.fromarray(np.random.default_rng(3).integers(0, 256, (16, 16, 4), dtype=np.uint8), "CMYK")
    path = tmp_path / "cmyk.jpg"
    im.save(path, format="JPEG")
    out = np.empty((16, 16, 4), np.uint8)
    assert img.srx_image_load(str(path).encode(), out.ctypes.data, out.nbytes) != 0
    assert b"CMYK" in img.srx_image_last_error()


def _gif(tmp_path, name, w=37, h=23, colors=19, **kw):
    rng = np.random.default_rng(31)
    a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    a[:6, :11] = a[0, 0]
    im = Image.fromarray(a, "RGB").quantize(colors)
    path = tmp_path / name
    im.save(path, format="GIF", **kw)
    return path


@pytest.mark.parametrize("kw", [{}, {"interlace": 1}, {"interlace": 0}, {"transparency": 3}, {"transparency": 0, "interlace": 1},
                                {"colors": 2}, {"colors": 256, "w": 130, "h": 70}])
def test_gif_against_the_reference_decoder(img, oracle, tmp_path, kw):
    """The first image of a GIF as RGBA: palette sizes, interlacing, a transparent index."""
    kw = dict(kw)
    size = {k: kw.pop(k) for k in ("w", "h", "colors") if k in kw}
    path = _gif(tmp_path, "t.gif", **size, **kw)
    got, ch, req = load(img, path)
    assert (ch, req) == (4, 4)
    pil = np.asarray(Image.open(path).convert("RGBA"))
    opaque = got[..., 3] == 255
    assert np.array_equal(got[opaque], pil[opaque]) and (got[~opaque] == 0).all()
    if "transparency" in kw:
        assert (~opaque).any()
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)


def test_gif_smaller_than_its_screen(img, oracle, tmp_path):
    """An image that does not cover the logical screen, with a background index: the pixels outside
    its rectangle get the background palette entry the way the reference's decoder writes it."""
    import struct
    path = _gif(tmp_path, "s.gif", w=20, h=12, colors=8, interlace=0)
    d = bytearray(path.read_bytes())
    d[6:10] = struct.pack("<HH", 26, 17)       # logical screen larger than the image
    d[11] = 5                                   # background colour index
    at = d.index(b"\x2c", 13 + 3 * (2 << (d[10] & 7)))
    d[at + 1:at + 5] = struct.pack("<HH", 4, 3)
    path.write_bytes(d)
    got, ch, req = load(img, path)
    assert got.shape == (17, 26, 4)
    inside = np.zeros((17, 26), bool)
    inside[3:15, 4:24] = True
    assert (got[~inside] == got[0, 0]).all() and got[0, 0, 3] == 255
    R = oracle.ref()
    if R is not None:
        ref, c, r = ref_load(R, path)
        assert (c, r) == (ch, req) and np.array_equal(ref, got)


def test_damaged_gifs_decode_as_the_reference_decoder_does(img, oracle, tmp_path):
    """Truncated files and files with flipped bytes in the LZW data: accepted or refused as the reference's loader
    does, and when accepted, with the same pixels (a raster cut short keeps what was decoded)."""
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref is not built here")
    rng = np.random.default_rng(77)
    accepted = refused = 0
    for k, kw in enumerate([dict(interlace=0), dict(interlace=1), dict(transparency=2, interlace=0), dict(colors=200, w=64, h=40, interlace=0)]):
        good = _gif(tmp_path, f"g{k}.gif", **kw).read_bytes()
        start = good.index(b"\x2c", 13)                     # image descriptor: damage only what follows it
        for trial in range(60):
            d = bytearray(good)
            if trial % 3 == 0:
                d = d[:int(rng.integers(start + 11, len(d)))]
            else:
                for _ in range(int(rng.integers(1, 4))):
                    d[int(rng.integers(start + 10, len(d)))] = int(rng.integers(0, 256))
            path = tmp_path / "damaged.gif"
            path.write_bytes(bytes(d))
            w, h, c, r = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            p = R.ref_load_image(str(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(r))
            out = np.zeros((h.value or 1, w.value or 1, 4), np.uint8)
            rc = img.srx_image_load(str(path).encode(), out.ctypes.data, out.nbytes) if p else img.srx_image_load(
                str(path).encode(), np.zeros(1 << 20, np.uint8).ctypes.data, 1 << 20)
            if not p:
                assert rc != 0, (k, trial)
                refused += 1
                continue
            ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, r.value)).copy()
            R.ref_free(p)
            assert rc == 0 and np.array_equal(out, ref), (k, trial, img.srx_image_last_error())
            accepted += 1
    assert accepted > 40 and refused > 10, (accepted, refused)
