"""GPU tests of the headless host program (superresolution_amd/srx_main): the SuperResolutionX /
SuperResolution classes driven through ParseCommandLineArgs -> OnInit -> {OnUpdate, OnRender}* ->
OnDestroy, as the reference's window loop drives them."""
import glob
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import superresolution_amd as S
from conftest import ASSETS, GOLDEN, ROOT, load_png_rgba

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "superresolution_amd", "srx_main")


def run(args, cwd):
    return subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


@pytest.fixture
def workdir(tmp_path):
    os.symlink(ASSETS, tmp_path / "Assets")       # weights and images are found relative to the CWD
    return tmp_path


def test_default_image_and_screenshot(workdir):
    """No arguments: Assets/Sashimi128.png, and the F11 path writes SuperResolutionX_<timestamp>.png (RGB)."""
    r = run([], workdir)
    assert r.returncode == 0, r.stderr
    assert "Number of weight tensors: 19" in r.stdout and "Loaded tensor: conv1/weights -> 2400" in r.stdout
    assert "output 256x256" in r.stdout
    shots = glob.glob(str(workdir / "SuperResolutionX_*.png"))
    assert len(shots) == 1
    im = Image.open(shots[0])
    assert im.mode == "RGB" and im.size == (256, 256)
    got = np.asarray(im)
    with S.SuperResolutionContext(128, 128) as ctx:
        want = ctx.run(load_png_rgba(os.path.join(ASSETS, "Sashimi128.png")))
    assert np.array_equal(got, want[..., :3])
    ref = load_png_rgba(os.path.join(GOLDEN, "sashimi128_phases_out.png"))
    assert np.abs(got.astype(int) - ref[..., :3].astype(int)).max() <= 1


@pytest.mark.parametrize("flag", ["-i", "/IMAGE", "-Image"])
def test_image_switch_spellings(workdir, flag):
    r = run([flag, "Assets/xiaoyao.png", "-o", "out.png", "-uma"], workdir)
    assert r.returncode == 0, r.stderr
    assert "output 182x182" in r.stdout
    got = np.asarray(Image.open(workdir / "out.png"))
    ref = load_png_rgba(os.path.join(GOLDEN, "xiaoyao_phases_out.png"))
    assert got.shape == (182, 182, 3)
    assert np.abs(got.astype(int) - ref[..., :3].astype(int)).max() <= 1


def test_update_mode_reruns_inference(workdir):
    r = run(["-i", "Assets/Sashimi.png", "-update", "-frames", "8", "-o", "o.png"], workdir)
    assert r.returncode == 0, r.stderr
    assert np.asarray(Image.open(workdir / "o.png")).shape == (512, 512, 3)


def test_failures_are_loud(workdir, tmp_path):
    r = run(["-warp"], workdir)
    assert r.returncode != 0 and "no software adapter" in r.stderr
    r = run(["-i", "Assets/missing.png"], workdir)
    assert r.returncode != 0 and "Unable to load image" in r.stderr
    empty = tmp_path / "empty"
    empty.mkdir()
    os.makedirs(empty / "Assets")
    os.symlink(os.path.join(ASSETS, "Sashimi128.png"), empty / "Assets" / "Sashimi128.png")
    r = run([], empty)                         # weights.bin missing -> loadWeights
    assert r.returncode != 0 and "Unable to open weight file" in r.stderr and "loadWeights" in r.stderr


@pytest.mark.parametrize("switches", [["-fp32"], ["-fp32", "-nhwc"], ["-nhwc"]])
def test_fp32_and_nhwc_switches(workdir, switches):
    """-fp32 (no fp16 support) and -nhwc (the vendor for which the reference picks NHWC) run the graph
    of generic layers; the picture stays within 1 LSB of the fused fp16 path."""
    r = run(["-i", "Assets/Sashimi128.png", "-o", "out.png"] + switches, workdir)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(workdir / "out.png").convert("RGB")).astype(int)
    with S.SuperResolutionContext(128, 128) as ctx:
        fast = ctx.run(load_png_rgba(os.path.join(ASSETS, "Sashimi128.png")))[..., :3].astype(int)
    assert (np.abs(got - fast) <= 1).mean() >= 0.999


@pytest.mark.parametrize("ext,kw", [("jpg", {"quality": 90, "subsampling": 2}), ("jpg", {"progressive": True}), ("bmp", {}), ("tga", {})])
def test_other_input_formats(workdir, ext, kw):
    """-i with the other formats the reference's loader accepts: the program upscales what the loader
    decodes (the decoders themselves are pinned against the reference's in tests/test_image_io.py)."""
    import ctypes as C
    src = Image.open(os.path.join(ASSETS, "Sashimi128.png")).convert("RGB")
    path = workdir / f"in.{ext}"
    src.save(path, **kw)
    r = run(["-i", str(path), "-o", "o.png"], workdir)
    assert r.returncode == 0, r.stderr
    assert "output 256x256" in r.stdout
    L = C.CDLL(os.path.join(ROOT, "superresolution_amd", "libsrximage.so"))
    L.srx_image_load.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    rgba = np.empty((128, 128, 4), np.uint8)
    assert L.srx_image_load(str(path).encode(), rgba.ctypes.data, rgba.nbytes) == 0
    with S.SuperResolutionContext(128, 128) as ctx:
        want = ctx.run(rgba)
    assert np.array_equal(np.asarray(Image.open(workdir / "o.png")), want[..., :3])


def test_devices_switch_shards_frames_over_instances(workdir):
    """-devices: one SuperResolutionX per listed device on its own host thread, frames dealt round-robin.
    On a one-GPU box the device is listed twice, so two instances (two ctxs, two streams, two threads)
    share it; both must produce the picture of a plain run."""
    r = run(["-i", "Assets/Sashimi.png", "-o", "one.png"], workdir)
    assert r.returncode == 0, r.stderr
    r = run(["-i", "Assets/Sashimi.png", "-devices", "0,0", "-frames", "7", "-o", "multi.png"], workdir)
    assert r.returncode == 0, r.stderr
    assert "instance 0 on device 0: 4 frames" in r.stdout and "instance 1 on device 0: 3 frames" in r.stdout
    assert "7 frames on 2 instances" in r.stdout
    want = np.asarray(Image.open(workdir / "one.png"))
    for k in range(2):
        assert np.array_equal(np.asarray(Image.open(workdir / f"multi.{k}.png")), want)
    r = run(["-devices", "0,99", "-frames", "2"], workdir)
    assert r.returncode != 0 and "device index out of range" in r.stderr


def test_more_instances_than_frames_and_output_names(workdir):
    """-devices with fewer frames than instances: the surplus instance renders and saves nothing; the index goes
    in front of the extension of the file NAME, wherever the directory has dots."""
    os.makedirs(workdir / "out.d")
    r = run(["-i", "Assets/Sashimi128.png", "-devices", "0,0", "-frames", "1", "-o", "./out.d/name"], workdir)
    assert r.returncode == 0, r.stderr
    assert "instance 1 on device 0: no frames" in r.stdout and "1 frames on 2 instances" in r.stdout
    assert os.path.exists(workdir / "out.d" / "name.0") and not os.path.exists(workdir / "out.d" / "name.1")
    assert not glob.glob(str(workdir / ".0*"))


def _stream_rate(stdout):
    import re
    m = re.search(r"= ([0-9.]+) frames/s host to host .*; ([0-9.]+) frames/s between", stdout)
    assert m, stdout
    return float(m.group(1)), float(m.group(2))


def test_stream_mode_small_image_equals_single_frame_path(workdir):
    """-stream: slots of frames in flight through the frame pipe (three slots, here four frames each); the
    picture is the one the single-frame path renders."""
    r = run(["-i", "Assets/Sashimi.png", "-o", "one.png"], workdir)
    assert r.returncode == 0, r.stderr
    r = run(["-i", "Assets/Sashimi.png", "-stream", "42", "-slot", "4", "-o", "s.png"], workdir)
    assert r.returncode == 0, r.stderr
    assert "stream on device 0: 44 frames" in r.stdout            # whole slots: 11 x 4
    assert np.array_equal(np.asarray(Image.open(workdir / "s.png")), np.asarray(Image.open(workdir / "one.png")))
    # F11 in stream mode: the shot is the first frame of a slot that has arrived
    before = set(glob.glob(str(workdir / "SuperResolutionX_*.png")))
    r = run(["-i", "Assets/Sashimi.png", "-stream", "24", "-slot", "4", "-shot"], workdir)
    assert r.returncode == 0, r.stderr
    shots = sorted(set(glob.glob(str(workdir / "SuperResolutionX_*.png"))) - before)
    assert len(shots) == 1, (shots, r.stdout)
    assert np.array_equal(np.asarray(Image.open(shots[0])), np.asarray(Image.open(workdir / "one.png")))
    os.remove(shots[0])
    # the generic-layer graphs have no frame pipe
    r = run(["-i", "Assets/Sashimi.png", "-stream", "8", "-fp32"], workdir)
    assert r.returncode != 0 and "fused fp16 kernels only" in r.stderr
    # two instances share the device, each with its own three slots in flight
    r = run(["-i", "Assets/Sashimi.png", "-stream", "64", "-devices", "0,0", "-o", "m.png"], workdir)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("stream on device 0: 32 frames") == 2 and "64 frames on 2 instances" in r.stdout
    for k in range(2):
        assert np.array_equal(np.asarray(Image.open(workdir / f"m.{k}.png")), np.asarray(Image.open(workdir / "one.png")))


def test_stream_mode_full_hd(workdir):
    """The throughput configuration from the C++ host: 1080p frames from page-locked host memory through a pass
    of eight frames per slot, three slots in flight, back to host memory.  The picture equals the library's
    single-frame result byte for byte; the rate is printed (and compared with bench.py's host-to-host leg in
    the design notes, not asserted here beyond being a plausible number)."""
    from conftest import synthetic_rgba
    rgba = synthetic_rgba(1920, 1080, 1080)
    Image.fromarray(rgba[..., :3]).save(workdir / "hd.png", compress_level=1)
    r = subprocess.run([EXE, "-i", "hd.png", "-stream", "960", "-o", "hd_out.png"], cwd=workdir, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "output 3840x2160" in r.stdout and "stream on device 0: 960 frames" in r.stdout
    whole, steady = _stream_rate(r.stdout)
    print(r.stdout.splitlines()[-2])
    assert whole > 100 and steady > 100
    with S.SuperResolutionContext(1920, 1080) as ctx:
        want = ctx.run(rgba)
    assert np.array_equal(np.asarray(Image.open(workdir / "hd_out.png")), want[..., :3])
