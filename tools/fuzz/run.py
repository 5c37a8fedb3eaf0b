#!/usr/bin/env python3
"""Mutation fuzzing of the host-side parsers (image decoders, weights.bin loader) under
AddressSanitizer + UndefinedBehaviorSanitizer (host code only; no GPU involved).

  python tools/fuzz/run.py [--runs 3000]

Builds the two harnesses beside this file into /tmp, mutates valid files (bit flips, deletions,
insertions, truncation) and reports every run that a sanitizer stopped."""
import argparse
import io
import os
import re
import subprocess
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HOST = os.path.join(ROOT, "superresolution_amd", "csrc", "host")
CSRC = os.path.join(ROOT, "superresolution_amd", "csrc")
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def build():
    subprocess.check_call(["g++"] + SAN + ["-I" + HOST, "-o", "/tmp/fuzz_decode_image", os.path.join(HERE, "decode_image.cpp")] +
                          [os.path.join(HOST, f) for f in ("ImageIO.cpp", "ImageFormats.cpp", "ImageJpeg.cpp", "ImageGif.cpp")] + ["-lz"])
    subprocess.check_call(["g++"] + SAN + ["-I" + CSRC, "-o", "/tmp/fuzz_load_weights", os.path.join(HERE, "load_weights.cpp"),
                           os.path.join(CSRC, "srx_weights.cpp")])


def mutate(rng, seed, hot=None):
    d = bytearray(seed)
    for _ in range(int(rng.integers(1, 8))):
        mode = int(rng.integers(0, 3))
        p = int(rng.integers(0, len(d)))
        if hot is not None and rng.random() < 0.6:
            p = min(len(d) - 1, max(0, int(rng.choice(hot)) + int(rng.integers(-12, 24))))
        if mode == 0 or hot is not None:
            d[p] = int(rng.integers(0, 256))
        elif mode == 1 and len(d) > 20:
            del d[p:p + int(rng.integers(1, 16))]
        else:
            d[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 8)), dtype=np.uint8))
    if rng.random() < 0.1:
        d = d[:int(rng.integers(1, len(d)))]
    return bytes(d)


def run(exe, path):
    try:
        r = subprocess.run([exe, path], capture_output=True, text=True, timeout=30)
        return r.returncode, r.stderr
    except subprocess.TimeoutExpired:
        return -99, "TIMEOUT"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3000)
    args = ap.parse_args()
    build()
    rng = np.random.default_rng(1)
    im = Image.fromarray(rng.integers(0, 256, (40, 37, 3), dtype=np.uint8), "RGB")
    seeds = {}
    for name, fmt, kw in (("s.png", "PNG", {}), ("s.jpg", "JPEG", {"quality": 80, "subsampling": 2}),
                          ("sp.jpg", "JPEG", {"progressive": True, "subsampling": 1}), ("sr.jpg", "JPEG", {"restart_marker_blocks": 2}),
                          ("s.gif", "GIF", {}), ("st.gif", "GIF", {"transparency": 2, "interlace": 1}), ("s.bmp", "BMP", {}), ("s.tga", "TGA", {}), ("sr.tga", "TGA", {"compression": "tga_rle"}), ("s.ppm", "PPM", {})):
        b = io.BytesIO()
        im.save(b, format=fmt, **kw)
        seeds[name] = b.getvalue()
    bad = 0
    names = list(seeds)
    for it in range(args.runs):
        name = names[it % len(names)]
        path = "/tmp/fuzz_case." + name.split(".")[1]
        open(path, "wb").write(mutate(rng, seeds[name]))
        rc, err = run("/tmp/fuzz_decode_image", path)
        if rc != 0:
            bad += 1
            print("image decoder stopped on a mutation of", name, "\n", err[-1500:])
    weights = open(os.path.join(ROOT, "superresolution_amd", "Assets", "weights.bin"), "rb").read()
    hot = [0] + [m.start() for m in re.finditer(rb"conv|scale|shift|bias|weights", weights)]
    for it in range(args.runs // 3):
        open("/tmp/fuzz_case.bin", "wb").write(mutate(rng, weights, hot))
        rc, err = run("/tmp/fuzz_load_weights", "/tmp/fuzz_case.bin")
        if rc != 0:
            bad += 1
            print("weights loader stopped\n", err[-1500:])
    print("runs:", args.runs + args.runs // 3, "stopped by a sanitizer:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
