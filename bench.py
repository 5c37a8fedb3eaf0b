#!/usr/bin/env python3
"""Benchmark of the MI355X super-resolution path: output megapixels/s.

  python bench.py --gpus N --steps K --warmup W [--config stream|tiles8k]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config stream (default): a step is one pass of the hot path (image->tensor, the seven convolutions
with fused upsample / residual add / UNORM conversion) over one batch of 64 synthetic 1920x1080 frames
per GPU that is already resident in HBM.  Every rank owns one GPU and its own batch.
--config tiles8k: a step is one synthetic 7680x4320 image, cut into eight bands with a 7-pixel halo
that are dealt to the ranks; afterwards rank 0 checks a tile against the untiled run, byte for byte.
Frames and tiles are independent: there is no collective on the data path.  Rank 0 prints one JSON line.

With --gpus N > 1 and no RANK in the environment this process starts the N rank processes itself
(torch.distributed.run) before it has touched torch or the GPU, relays rank 0's line and exits with
the launcher's status.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
W8K, H8K, TILES_X, TILES_Y = 7680, 4320, 1, 8     # eight bands of 540 rows: full-width tiles keep the strips whole
# FLOPs (2 * MAC, unpadded channels, borders counted as full) per LR pixel.  conv_up1 is counted
# in the reference's direct form (5x5 on the 2x-upsampled tensor: 100 products per input pixel, input and
# output channel) and in the form each kernel evaluates: the layer-wise kernels' four 3x3 phase filters (36
# products) and the fused kernel's row-shared form (columns folded per phase, rows shared between the two output
# rows: 30 products, 3.33x fewer than the direct form; same result up to the rounding of the folded taps).
UP1_DIRECT, UP1_PHASES, UP1_ROW_SHARED = 409600, 147456, 122880
FLOP_PER_LR_PIXEL = {"conv1": 4800, "conv2": 36864, "conv3": 73728, "conv_up1": UP1_PHASES,
                     "conv4": 73728, "conv5": 73728, "conv6_add_unorm": 6912,
                     "conv4_conv5_conv6_add_unorm": 73728 + 73728 + 6912,
                     "pack_conv1_conv2_conv3_conv_up1": 4800 + 36864 + 73728 + UP1_ROW_SHARED}
FLOP_PER_LR_PIXEL_DIRECT = dict(FLOP_PER_LR_PIXEL, conv_up1=UP1_DIRECT,
                                pack_conv1_conv2_conv3_conv_up1=4800 + 36864 + 73728 + UP1_DIRECT)
FLOP_PER_PIXEL = 4800 + 36864 + 73728 + UP1_ROW_SHARED + 73728 + 73728 + 6912     # whole path as the fused kernels evaluate it: 392640
FLOP_PER_PIXEL_DIRECT = 679360                                      # whole path, the reference's direct form
FLOP_PER_PIXEL_PHASES = 417216                                      # whole path with conv_up1 as four 3x3 phase filters
FLOP_PER_LR_PIXEL_PHASES = dict(FLOP_PER_LR_PIXEL, pack_conv1_conv2_conv3_conv_up1=4800 + 36864 + 73728 + UP1_PHASES)
PEAK_FP16_DENSE = 2.5e15                       # MI355X dense fp16 MFMA, FLOP/s
PEAK_HBM = 8.0e12


def synthetic_frame(seed, w=W, h=H):
    """Uniform noise blended 50/50 with a gradient, alpha 255; made in bands of rows to bound memory."""
    rng = np.random.default_rng(seed)
    out = np.empty((h, w, 4), np.uint8)
    out[..., 3] = 255
    for y0 in range(0, h, 1080):
        y1 = min(h, y0 + 1080)
        noise = rng.integers(0, 256, size=(y1 - y0, w, 3), dtype=np.int64)
        yy, xx = np.mgrid[y0:y1, 0:w]
        out[y0:y1, :, :3] = (noise + ((xx + yy) & 255)[..., None]) // 2
    return out


def cpu_baseline():
    """The CPU oracle (reference graph, direct form) timed on this host on a bounded sample.
    Returns the report and what the oracle computed for the whole frame (the 8-bit image and conv_up1's
    tensor), which parity_block() compares the GPU path with."""
    from oracle import sro
    host_cores = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, host_cores)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sro.build()
    import superresolution_amd as S
    model = sro.Model(sro.Weights(S.default_weights_path()), 0, 0)
    frame = synthetic_frame(1080)
    model.run(frame[:64, :64].copy())                      # touch code and pages
    t0 = time.perf_counter()
    ref_image = model.run(frame, keep_tensors=True)
    dt = time.perf_counter() - t0
    ref = {"frame": frame, "image": ref_image, "conv_up1": model.tensor(sro.T_CONV_UP1)}
    out = {"value": round(4 * W * H / 1e6 / dt, 4), "unit": "output megapixels/s", "cores": threads,
           "host_cores": host_cores, "kind": "port",
           "sample": f"one whole synthetic {W}x{H} frame through oracle/sr_oracle.c "
                     f"(reference graph, direct conv_up1), OpenMP {threads} of the host's {host_cores} "
                     f"logical CPUs, {dt:.2f} s"}
    # one thread, on a 512x512 crop of the same frame
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        crop = np.ascontiguousarray(frame[:512, :512])
        t0 = time.perf_counter()
        model.run(crop)
        dt1 = time.perf_counter() - t0
        out["single_thread"] = {"value": round(4 * 512 * 512 / 1e6 / dt1, 4), "unit": "output megapixels/s", "cores": 1,
                                "sample": f"512x512 crop of the same frame, one thread, {dt1:.2f} s"}
    except OSError:
        pass
    return out, ref


# the tolerance the path is specified to: 8-bit image within 1 LSB on at least 99.9 % of the pixels,
# tensors within 2e-2 (fp16 storage, fp32 accumulation in another order than the oracle's running sums)
PARITY_IMAGE_FRAC, PARITY_TENSOR_TOL = 0.999, 2e-2


def parity_block(S, ref, device):
    """The GPU path on the very frame the cpu_baseline leg pushed through the oracle, in the plan the
    headline number runs (a pass of eight 1080p frames: first and last slot), against the oracle's
    whole 3840x2160 image and its conv_up1 tensor (reference graph, direct 5x5 on the upsampled tensor)."""
    frame = ref["frame"]
    h, w = frame.shape[:2]
    with S.SuperResolutionContext(w, h, batch=8, device=device) as ctx:
        for i in (0, 7):
            ctx.upload_rgba8(i, frame)
        ctx.upscale()
        ctx.sync()
        images = [ctx.tensor_to_image(i) for i in (0, 7)]
        a4 = ctx.read_tensor(4, frame=7)
        plan = ctx.frames_per_launch
    d = np.abs(images[0][..., :3].astype(np.int16) - ref["image"][..., :3].astype(np.int16))
    a4_err = 0.0
    for c0 in range(0, a4.shape[0], 8):                       # in slices: bounds the float32 copies
        a4_err = max(a4_err, float(np.abs(a4[c0:c0 + 8].astype(np.float32) - ref["conv_up1"][c0:c0 + 8]).max()))
    out = {"within_1_lsb": round(float((d <= 1).mean()), 6), "identical": round(float((d == 0).mean()), 6),
           "max_lsb": int(d.max()), "a4_max_abs": round(a4_err, 6),
           "alpha_255": bool((images[0][..., 3] == 255).all()),
           "slots_equal": bool(np.array_equal(images[0], images[1])),
           "tolerance": {"within_1_lsb": PARITY_IMAGE_FRAC, "a4_max_abs": PARITY_TENSOR_TOL},
           "sample": f"the cpu_baseline frame ({w}x{h}, every one of its {4 * w * h} output pixels and the whole "
                     f"32-channel conv_up1 tensor) through a pass of {plan} frames, slots 0 and 7, against "
                     "oracle/sr_oracle.c (direct form)"}
    out["ok"] = bool(out["within_1_lsb"] >= PARITY_IMAGE_FRAC and out["a4_max_abs"] <= PARITY_TENSOR_TOL
                     and out["alpha_255"] and out["slots_equal"])
    return out


def pcie_inclusive(S, frames, device):
    """Frames that start and end in (pinned) host memory, through the frame pipe: 3 slots of 8 frames
    in flight.  Reported beside the headline number, never as `value`."""
    per_slot, slots, rounds = 8, 3, 8
    with S.SuperResolutionContext(W, H, batch=per_slot * slots, device=device) as ctx, \
            S.FramePipe(ctx, per_slot, in_channels=3, out_channels=3) as pipe:
        for s in range(slots):
            buf = pipe.input(s)
            for k in range(per_slot):
                buf[k] = frames[k % len(frames)][..., :3]
        # jobs 0, 1, 2, ... go to slots 0, 1, 2, 0, ...; a slot is resubmitted as soon as its previous job
        # has been waited for (a real caller consumes and refills the pinned buffers at that point).
        # The rate is taken between two completions in the steady state.
        jobs, first_timed = (rounds + 2) * slots, 2 * slots
        done_at = []
        for j in range(jobs + slots):
            s_ = j % slots
            if j >= slots:
                pipe.wait(s_)
                done_at.append(time.perf_counter())
            if j < jobs:
                pipe.submit(s_)
        dt = done_at[-1] - done_at[first_timed - 1]
        rounds = (len(done_at) - first_timed) // slots
    n = rounds * slots * per_slot
    return {"frames_per_s": round(n / dt, 1), "output_megapixels_per_s": round(n / dt * 4 * W * H / 1e6, 1),
            "note": f"{n} frames host->GPU->host through srx_pipe_* (pinned buffers, {slots} slots x {per_slot} frames, "
                    "upload / compute / download on three streams, RGB8 buffers read and written by the fused kernels themselves); "
                    "6.2 MB in + 24.9 MB out per frame over PCIe"}


def single_frame_latency(S, frame, device, reps=50):
    """The batch-1 configuration: one resident 1080p frame, upscale + synchronize, median of `reps`."""
    import torch
    with S.SuperResolutionContext(W, H, batch=1, device=device) as ctx:
        ctx.upload_rgba8(0, frame)
        stream = torch.cuda.current_stream().cuda_stream or None
        for _ in range(5):
            ctx.upscale(stream)
        torch.cuda.synchronize()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.upscale(stream)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    return round(sorted(times)[len(times) // 2] * 1e3, 4)


class DryContext:
    """Stands in for a ctx under --dry-run: no GPU, no kernel, no number.  It lets the launcher, the
    rendezvous, the barrier / max-over-ranks timing and the report be exercised on a machine without a GPU."""

    def __init__(self, width, height, batch=1, device=0):
        self.width, self.height, self.batch = width, height, batch
        self.frames_per_launch = min(batch, 8)

    def upload_rgba8(self, frame, rgba, stream=None):
        assert rgba.shape == (self.height, self.width, 4)

    def upscale(self, stream=None):
        time.sleep(1e-3)

    def stage_names(self):
        return []

    def close(self):
        pass


def launch_ranks(n, argv):
    """Parent of an N-rank run: starts the ranks with torch.distributed.run and relays rank 0's line.
    Runs before torch is imported: this process never initialises a GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exit status {proc.returncode})", file=sys.stderr)
        return proc.returncode
    if len(lines) != 1:
        print(f"bench.py: expected one result line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["stream", "tiles8k"], default="stream",
                    help="stream: 1080p frames, 64 per GPU per step (the headline); tiles8k: one 7680x4320 image in eight halo tiles")
    ap.add_argument("--batch", type=int, default=64, help="1080p frames per GPU per step (stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-reps", type=int, default=20)
    ap.add_argument("--no-pcie-leg", action="store_true", help="skip the host-to-host (PCIe inclusive) measurement")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the single-frame latency measurement")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU and no kernels: exercise launcher, rendezvous (gloo), timing protocol and report; value is null")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # the host driver of these boxes only supports dmabuf IPC: RCCL needs this before the runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = args.dry_run
    if os.environ.get("SRX_BENCH_FAIL_RANK") == str(rank):      # used by the launcher's test
        raise SystemExit(f"rank {rank}: failing on request")
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    n_dev = 0 if dry else torch.cuda.device_count()
    shared = dry or world > n_dev                 # more ranks than GPUs: a rehearsal of the N > 1 path on fewer GPUs
    if not dry:
        local_rank %= n_dev
        torch.cuda.set_device(local_rank)
    # A process group whenever a launcher started this rank (RANK set), also for one rank: a 1-GPU run
    # under torch.distributed.run then goes through the very calls of the 8-GPU run.
    grouped = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    backend = None
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:      # RCCL refuses two ranks on one GPU; the barrier and the max-reduce then go over gloo
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from superresolution_amd import sharding
    if dry:
        S, Ctx, stream = None, DryContext, None
    else:
        import superresolution_amd as S
        Ctx = S.SuperResolutionContext
        stream = torch.cuda.current_stream().cuda_stream or None

    # ---- the work of this rank, resident in HBM before the clock starts
    B = args.batch
    ctxs, distinct, mine, image8k = [], [], [], None
    if args.config == "stream":
        ctx = Ctx(W, H, batch=B, device=local_rank)
        # synthetic frames: a few distinct seeds per rank, repeated to fill the batch (values do not
        # change the timing of dense MFMA work; parity on this size is covered in tests/)
        distinct = [synthetic_frame(rank * B + i) for i in range(min(B, 4))]
        for i in range(B):
            ctx.upload_rgba8(i, distinct[i % len(distinct)])
        ctxs.append(ctx)
        pixels_per_step = B * W * H                       # input pixels this rank processes per step
    else:
        image8k = synthetic_frame(4320, W8K, H8K)         # every rank makes the same image and takes its own tiles
        mine = sharding.tiles_for_rank(sharding.tile_plan(W8K, H8K, TILES_X, TILES_Y), rank, world)
        for t in mine:
            ctx = Ctx(t.ex1 - t.ex0, t.ey1 - t.ey0, device=local_rank)
            ctx.upload_rgba8(0, np.ascontiguousarray(image8k[t.ey0:t.ey1, t.ex0:t.ex1]))
            ctxs.append(ctx)
        pixels_per_step = sum((t.x1 - t.x0) * (t.y1 - t.y0) for t in mine)   # interiors: halo rows are overhead

    def barrier():
        if grouped:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def step():
        for c in ctxs:
            c.upscale(stream)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if not dry:
        torch.cuda.synchronize()
    own = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(own)         # the job takes as long as its slowest rank
    per_rank_s = sharding.gather_over_ranks(own)
    per_rank_pixels = sharding.gather_over_ranks(float(pixels_per_step))
    barrier()

    # per-kernel durations, HIP events on the launch stream (one launch covers the frames of
    # one workspace pass: ctx.frames_per_launch, 8 for the 1080p stream)
    stage_ms = {}
    if rank == 0 and ctxs and not dry:
        ctx = ctxs[0]
        for i, name in enumerate(ctx.stage_names()):
            ctx.launch_stage(i, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.stage_reps):
                ctx.launch_stage(i, stream)
            e1.record()
            e1.synchronize()
            stage_ms[name] = e0.elapsed_time(e1) / args.stage_reps
    barrier()

    tile_check = None
    if args.config == "tiles8k" and rank == 0 and not dry and mine:
        # one tile against the untiled run on the same GPU: its interior, which includes a seam, byte for byte
        t = mine[0]
        tile_out = t.crop(ctxs[0].tensor_to_image(0))
        with Ctx(W8K, H8K, device=local_rank) as whole:
            full = whole.run(image8k)
        if not np.array_equal(tile_out, full[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1]):
            raise SystemExit("tiles8k: the tile's interior differs from the untiled run")
        tile_check = (f"interior of tile 0 ({2 * (t.x1 - t.x0)}x{2 * (t.y1 - t.y0)} output pixels, rows {2 * t.y0}..{2 * t.y1}) "
                      f"is byte-identical to the same window of the untiled {W8K}x{H8K} run")
        del full

    if rank == 0:
        total_pixels = sum(per_rank_pixels)                       # input pixels per step over all ranks
        value = None if dry else 4 * total_pixels * args.steps / elapsed / 1e6
        out = {
            "metric": "output megapixels/sec (1080p->4K 2x)" if args.config == "stream"
                      else "output megapixels/sec (7680x4320 -> 15360x8640, 8 halo tiles)",
            "value": None if value is None else round(value, 2),
            "unit": "output megapixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if args.config == "stream" else "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
        }
        if args.config == "stream":
            out["config"] = {"workload": f"synthetic 1920x1080 RGB frame stream -> 3840x2160, {B} frames per GPU "
                                         f"per step, partitioned by frame, bundled weights.bin",
                             "frames_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world}, no collective"}
            if not dry:
                out["frames_per_s"] = round(B * world * args.steps / elapsed, 2)
                out["ms_per_frame_per_gpu"] = round(elapsed / args.steps / B * 1e3, 4)
                out["per_rank_frames_per_s"] = [round(B * args.steps / s, 2) for s in per_rank_s]
        else:
            out["config"] = {"workload": f"one synthetic {W8K}x{H8K} RGB image -> {2 * W8K}x{2 * H8K} per step, "
                                         f"{TILES_X * TILES_Y} spatial tiles with a {sharding.HALO}-pixel halo dealt to the ranks, "
                                         "bundled weights.bin",
                             "tiles_per_gpu": len(mine), "halo": sharding.HALO,
                             "parallelism": f"tile-sharded x{world}, no collective"}
            if not dry:
                out["per_rank_output_megapixels_per_s"] = [round(4 * p * args.steps / s / 1e6, 2)
                                                           for p, s in zip(per_rank_pixels, per_rank_s)]
            out["tile_check"] = tile_check
        out["per_rank_seconds"] = [round(s, 5) for s in per_rank_s]
        out["timing_collectives"] = backend        # "nccl" (RCCL), "gloo" (rehearsal) or None (single process, no group)
        if not dry:
            per_gpu = total_pixels * args.steps / elapsed / world      # input pixels per second per GPU
            out["mfma_frac_whole_path"] = round(per_gpu * FLOP_PER_PIXEL / PEAK_FP16_DENSE, 4)
            out["mfma_frac_whole_path_direct_form"] = round(per_gpu * FLOP_PER_PIXEL_DIRECT / PEAK_FP16_DENSE, 4)
            # the same throughput priced as if conv_up1 were still evaluated as four 3x3 phase filters (36 products
            # instead of 30): the figure earlier measurements of this path quoted
            out["mfma_frac_whole_path_four_phase_form"] = round(per_gpu * FLOP_PER_PIXEL_PHASES / PEAK_FP16_DENSE, 4)
            ctx = ctxs[0]
            frames_per_launch = ctx.frames_per_launch      # frames one launch of either fused kernel covers
            out["frames_per_launch"] = frames_per_launch
            out["stage_ms"] = {k: round(v, 4) for k, v in stage_ms.items()}
            # dominant kernel = the conv stage with the longest launch
            dom = max((n for n in stage_ms if n in FLOP_PER_LR_PIXEL), key=lambda n: stage_ms[n])
            dom_s = stage_ms[dom] / 1e3
            launch_pixels = ctx.width * ctx.height * frames_per_launch
            achieved = FLOP_PER_LR_PIXEL[dom] * launch_pixels / dom_s / 1e12
            achieved_direct = FLOP_PER_LR_PIXEL_DIRECT[dom] * launch_pixels / dom_s / 1e12
            traffic, traffic_note, tj = None, "traffic: none committed for this configuration", {}
            tp = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tp):
                try:
                    tj = json.load(open(tp))
                    entry = tj if args.config == "stream" else tj.get("tiles8k", {})
                    traffic = entry.get(dom)
                    if traffic is not None:
                        traffic_note = ("traffic is NOT measured by this run: it is the HBM byte count per launch from the "
                                        "committed rocprofv3 --pmc passes (profiles/traffic.json, " + entry.get("_measured", "date not recorded") +
                                        ("" if args.config == "stream" else "; " + entry.get("_config", "")) + ")")
                except Exception:
                    traffic, tj = None, {}
            # what the matrix units actually run at: the MFMA work the dominant kernel issues per launch, padding
            # included (the library reports it from its own plan), over the same launch time
            issued = ctx.stage_issued_flops(ctx.stage_names().index(dom))
            out["roofline"] = {
                "bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP16_DENSE / 1e12,
                "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_FP16_DENSE, 4), "traffic": traffic,
                "achieved_direct_form": round(achieved_direct, 2),
                "frac_direct_form": round(achieved_direct * 1e12 / PEAK_FP16_DENSE, 4),
                "frac_four_phase_form": round(FLOP_PER_LR_PIXEL_PHASES[dom] * launch_pixels / dom_s / PEAK_FP16_DENSE, 4),
                "issued": round(issued / dom_s / 1e12, 2), "frac_issued": round(issued / dom_s / PEAK_FP16_DENSE, 4),
                "note": "achieved = FLOPs of the kernel's algorithm (conv_up1 in its row-shared form on the low-resolution "
                        "tensor, 3.33x fewer MACs than the reference's direct 5x5-on-upsampled form and 1.2x fewer than "
                        "four 3x3 phase filters: a cheaper form lowers this figure at equal speed) x input pixels per "
                        "launch / HIP-event launch time; *_direct_form counts conv_up1 as the "
                        "reference evaluates it; issued = the MFMA work the kernel hands to the matrix units per launch, "
                        "padded columns, taps and warm-up rows included (srx_stage_issued_flops).  The chip holds about 1.8 GHz (not 2.4) under this MFMA "
                        "load: a bare LDS-fed MFMA loop of the same shape measures 1.70 PFLOP/s here "
                        "(tools/mfma_shape_bench.hip).  " + traffic_note}
            info = ctx.device_info()
            out["device"] = {"arch": info["arch"], "compute_units": info["compute_units"], "max_clock_mhz": info["clock_khz"] // 1000,
                             "peak_note": "2.5 PFLOP/s = 256 CUs x 4096 fp16 FLOP/clk x 2.4 GHz"}
        if shared:
            out["rehearsal"] = ("dry run: no GPU work, no number" if dry else
                                f"{world} ranks on {n_dev} GPU(s): not a scaling measurement")
        if not dry and world == 1 and args.config == "stream":
            if not args.no_latency_leg:
                out["latency_ms_single_frame"] = single_frame_latency(S, distinct[0], local_rank)
                lat = tj.get("latency", {}) if not dry else {}
                if all(k in lat for k in FLOP_PER_LR_PIXEL if k.startswith(("pack_", "conv4_conv5"))):
                    out["latency_hbm_bytes_per_frame"] = {"pack_conv1_conv2_conv3_conv_up1": lat["pack_conv1_conv2_conv3_conv_up1"],
                                                          "conv4_conv5_conv6_add_unorm": lat["conv4_conv5_conv6_add_unorm"],
                                                          "note": "committed rocprofv3 --pmc passes of the batch-1 plan, not measured by this run (" + lat.get("_measured", "") + ")"}
            if not args.no_pcie_leg:
                out["pcie_inclusive"] = pcie_inclusive(S, distinct, local_rank)
        parity_failed = False
        if not dry and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline()
            if args.config == "stream":
                out["parity"] = parity_block(S, ref, local_rank)
                parity_failed = not out["parity"]["ok"]
            del ref
        print(json.dumps(out), flush=True)
        if parity_failed:
            print("bench.py: the GPU path is outside its tolerance against the oracle (see \"parity\")", file=sys.stderr)
            for c in ctxs:
                c.close()
            raise SystemExit(3)
    for c in ctxs:
        c.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
