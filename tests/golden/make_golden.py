"""Generates the committed golden fixtures from the CPU oracle.

The reference ships no golden vectors for this path (it has no tests), so the
fixtures are produced here by oracle/sr_oracle.c after it has been pinned:
  * its fp16 truncation is compared bit-for-bit with the reference's own
    Float16Compressor.h compiled from where it lies (oracle/_ref), and
  * its graph is cross-checked against torch-CPU float64 (oracle/crosscheck_torch.py).
Only data is written: inputs (seeded synthetic) and expected outputs.

Run from the repository root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sro  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(ROOT, "superresolution_amd", "Assets")


def synthetic_rgba(w, h, seed):
    """uniform noise blended 50/50 with a smooth gradient, alpha 255"""
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, size=(h, w, 3), dtype=np.int64)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx + yy) & 255)[..., None]
    rgb = ((noise + grad) // 2).astype(np.uint8)
    return np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)


def windows(h, w, size=16):
    s = min(size, h, w)
    return {"tl": (0, 0, s), "br": (h - s, w - s, s), "c": ((h - s) // 2, (w - s) // 2, s)}


def upsample_fixture():
    """upsample_linear.npz: inputs and expected outputs of the generic upsample layer's LINEAR mode, from an
    implementation that is not ours: torch-CPU float64 interpolate(mode="bilinear", align_corners=False), i.e.
    half-pixel centres with the sample position clamped at the edges.  That DirectML's UPSAMPLE_2D LINEAR uses
    this rule is taken from its documentation and cannot be run here: parity with DirectML stays unpinned."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(20261111)
    out = {}
    for k, (n, c, h, w, sx, sy) in enumerate([(2, 3, 5, 7, 3, 2), (1, 8, 9, 4, 2, 2), (1, 1, 1, 6, 4, 1), (1, 5, 6, 1, 1, 3), (1, 16, 11, 13, 2, 4)]):
        x = rng.uniform(-1, 1, (n, c, h, w)).astype(np.float16)            # representable in both storage types
        y = F.interpolate(torch.from_numpy(x.astype(np.float64)), scale_factor=(sy, sx), mode="bilinear", align_corners=False)
        out[f"x{k}"] = x
        out[f"scale{k}"] = np.array([sx, sy], np.int32)
        out[f"y{k}"] = y.numpy()
    np.savez_compressed(os.path.join(OUT, "upsample_linear.npz"), **out)
    print("upsample_linear.npz")


def main():
    if "--only-upsample" in sys.argv:
        return upsample_fixture()
    upsample_fixture()
    sro.build()
    W = sro.Weights(os.path.join(ASSETS, "weights.bin"))

    # ---- folded weights, pinned against the reference compressor -------------
    R = sro.ref()
    raw = W.tensors()
    folded = {}
    for li, (conv, scale, shift, sizes) in enumerate(sro.LAYERS):
        f, b = W.create_weight_tensors(conv, scale, shift, sizes)
        if R is not None:
            w32 = raw[conv].reshape(sizes[0], -1)
            if scale:
                w32 = (raw[scale][:, None] * w32).astype(np.float32)
            ref = np.empty(w32.size, np.uint16)
            w32 = np.ascontiguousarray(w32, np.float32)
            R.ref_f16_compress_array(w32.ctypes.data, ref.ctypes.data, w32.size)
            assert np.array_equal(ref, f), conv
            if b is not None:
                sh = np.ascontiguousarray(raw[shift], np.float32)
                refb = np.empty(sh.size, np.uint16)
                R.ref_f16_compress_array(sh.ctypes.data, refb.ctypes.data, sh.size)
                assert np.array_equal(refb, b), shift
        folded[f"filter{li}"] = f
        if b is not None:
            folded[f"bias{li}"] = b
    np.savez_compressed(os.path.join(OUT, "weights_fp16.npz"), **folded)
    print("weights_fp16.npz", "(checked against reference Float16Compressor)" if R else "(reference absent)")

    # ---- images --------------------------------------------------------------
    cases = {
        "sashimi128": sro.load_rgba(os.path.join(ASSETS, "Sashimi128.png")),
        "sashimi": sro.load_rgba(os.path.join(ASSETS, "Sashimi.png")),
        "xiaoyao": sro.load_rgba(os.path.join(ASSETS, "xiaoyao.png")),
        "synthetic_64x48": synthetic_rgba(64, 48, 6448),
        "synthetic_33x21": synthetic_rgba(33, 21, 3321),
    }
    manifest = []
    for name, rgba in cases.items():
        for mode, tag in ((0, "direct"), (1, "phases")):
            m = sro.Model(W, 0, mode)
            out = m.run(rgba, keep_tensors=True)
            crops = {}
            for tid, tname in enumerate(sro.TENSOR_NAMES):
                t = m.tensor(tid).astype(np.float16)
                digest = hashlib.sha256(t.tobytes()).hexdigest()
                manifest.append(f"{name} {tag} {tname} {t.shape[0]}x{t.shape[1]}x{t.shape[2]} {digest}")
                for wname, (y, x, s) in windows(t.shape[1], t.shape[2]).items():
                    crops[f"{tname}_{wname}"] = t[:, y:y + s, x:x + s]
            np.savez_compressed(os.path.join(OUT, f"{name}_{tag}_crops.npz"), **crops)
            Image.fromarray(out).save(os.path.join(OUT, f"{name}_{tag}_out.png"), optimize=True)
            manifest.append(f"{name} {tag} image {out.shape[1]}x{out.shape[0]} {hashlib.sha256(out.tobytes()).hexdigest()}")
        if name.startswith("synthetic"):
            Image.fromarray(rgba).save(os.path.join(OUT, f"{name}_in.png"), optimize=True)
        print(name, rgba.shape)
    with open(os.path.join(OUT, "MANIFEST.txt"), "w") as f:
        f.write("\n".join(manifest) + "\n")


if __name__ == "__main__":
    main()
