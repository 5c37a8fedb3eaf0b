#!/usr/bin/env python3
"""Benchmark of the MI355X super-resolution path: output megapixels/s for 1080p -> 4K.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (image->tensor, the seven convolutions with fused
upsample / residual add / UNORM conversion) over one batch of synthetic 1920x1080 frames that
is already resident in HBM.  Every rank owns one GPU and its own batch (frames are
independent: no collective on the data path); rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
OUT_MP = 4 * W * H / 1e6                       # 8.2944 output megapixels per frame
# FLOPs (2 * MAC, unpadded channels, borders counted as full) per LR pixel.  conv_up1 is counted
# twice: in the reference's direct form (5x5 on the 2x-upsampled tensor) and in the four-phase 3x3
# form the kernels evaluate (2.78x fewer MACs, same result up to the rounding of the summed taps).
UP1_DIRECT, UP1_PHASES = 409600, 147456
FLOP_PER_LR_PIXEL = {"conv1": 4800, "conv2": 36864, "conv3": 73728, "conv_up1": UP1_PHASES,
                     "conv4": 73728, "conv5": 73728, "conv6_add_unorm": 6912,
                     "conv4_conv5_conv6_add_unorm": 73728 + 73728 + 6912,
                     "pack_conv1_conv2_conv3_conv_up1": 4800 + 36864 + 73728 + UP1_PHASES}
FLOP_PER_LR_PIXEL_DIRECT = dict(FLOP_PER_LR_PIXEL, conv_up1=UP1_DIRECT,
                                pack_conv1_conv2_conv3_conv_up1=4800 + 36864 + 73728 + UP1_DIRECT)
FLOP_PER_FRAME = 417216 * W * H                                     # 865.1 GFLOP, four-phase form
FLOP_PER_FRAME_DIRECT = 679360 * W * H                              # 1408.72 GFLOP, the reference's direct form
PEAK_FP16_DENSE = 2.5e15                       # MI355X dense fp16 MFMA, FLOP/s
PEAK_HBM = 8.0e12


def synthetic_frame(seed):
    rng = np.random.default_rng(seed)
    noise = rng.integers(0, 256, size=(H, W, 3), dtype=np.int64)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb = ((noise + ((xx + yy) & 255)[..., None]) // 2).astype(np.uint8)
    return np.concatenate([rgb, np.full((H, W, 1), 255, np.uint8)], axis=2)


def cpu_baseline():
    """The CPU oracle (reference graph, direct form) timed on this host on a bounded sample."""
    from oracle import sro
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    sro.build()
    import superresolution_amd as S
    model = sro.Model(sro.Weights(S.default_weights_path()), 0, 0)
    cw, ch = W, H
    crop = synthetic_frame(1080)
    model.run(crop[:64, :64].copy())                       # touch code and pages
    t0 = time.perf_counter()
    model.run(crop)
    dt = time.perf_counter() - t0
    out = {"value": round(4 * cw * ch / 1e6 / dt, 4), "unit": "output megapixels/s", "cores": threads,
           "kind": "port",
           "sample": f"one whole synthetic {cw}x{ch} frame through oracle/sr_oracle.c "
                     f"(reference graph, direct conv_up1), OpenMP {threads} threads, {dt:.2f} s"}
    # the reference's own CPU-sized case on one thread: the bundled 128x128 image
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
        small = sro.load_rgba(os.path.join(os.path.dirname(S.default_weights_path()), "Sashimi128.png"))
        t0 = time.perf_counter()
        model.run(small)
        dt1 = time.perf_counter() - t0
        out["single_thread"] = {"value": round(4 * 128 * 128 / 1e6 / dt1, 4), "unit": "output megapixels/s",
                                "sample": f"Sashimi128.png (128x128), one thread, {dt1:.2f} s"}
    except OSError:
        pass
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
    return {"frames_per_s": round(n / dt, 1), "output_megapixels_per_s": round(n / dt * OUT_MP, 1),
            "note": f"{n} frames host->GPU->host through srx_pipe_* (pinned buffers, {slots} slots x {per_slot} frames, "
                    "upload / compute / download on three streams, RGB8 buffers); 6.2 MB in + 24.9 MB out per frame over PCIe"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="1080p frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stage-reps", type=int, default=20)
    ap.add_argument("--no-pcie-leg", action="store_true", help="skip the host-to-host (PCIe inclusive) measurement")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the single-frame latency measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    shared = world > torch.cuda.device_count()    # more ranks than GPUs: a rehearsal of the N > 1 path on fewer GPUs
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:      # RCCL refuses two ranks on one GPU; the barrier and the max-reduce then go over gloo
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import superresolution_amd as S
    B = args.batch
    stream = torch.cuda.current_stream().cuda_stream or None
    ctx = S.SuperResolutionContext(W, H, batch=B, device=local_rank)
    # synthetic frames: a few distinct seeds per rank, repeated to fill the batch (values do not
    # change the timing of dense MFMA work; parity on this size is covered in tests/)
    distinct = [synthetic_frame(rank * B + i) for i in range(min(B, 4))]
    for i in range(B):
        ctx.upload_rgba8(i, distinct[i % len(distinct)])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.upscale(stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.upscale(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    from superresolution_amd.sharding import max_over_ranks
    elapsed = max_over_ranks(elapsed)          # the job takes as long as its slowest rank
    barrier()

    # per-kernel durations, HIP events on the launch stream (one launch covers the frames of
    # one workspace pass: ctx.frames_per_launch, 8 at this size)
    names = ctx.stage_names()
    stage_ms = {}
    if rank == 0:
        for i, name in enumerate(names):
            ctx.launch_stage(i, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.stage_reps):
                ctx.launch_stage(i, stream)
            e1.record()
            e1.synchronize()
            stage_ms[name] = e0.elapsed_time(e1) / args.stage_reps
    barrier()

    if rank == 0:
        frames = B * world * args.steps
        fps = frames / elapsed
        value = fps * OUT_MP
        frames_per_launch = ctx.frames_per_launch      # frames one launch of either fused kernel covers
        # dominant kernel = the conv stage with the longest launch
        dom = max((n for n in stage_ms if n in FLOP_PER_LR_PIXEL), key=lambda n: stage_ms[n])
        dom_s = stage_ms[dom] / 1e3
        dom_flops = FLOP_PER_LR_PIXEL[dom] * W * H * frames_per_launch
        achieved = dom_flops / dom_s / 1e12
        achieved_direct = FLOP_PER_LR_PIXEL_DIRECT[dom] * W * H * frames_per_launch / dom_s / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "output megapixels/sec (1080p->4K 2x)",
            "value": round(value, 2),
            "unit": "output megapixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"synthetic 1920x1080 RGB frame stream -> 3840x2160, {B} frames per GPU "
                                   f"per step, partitioned by frame, bundled weights.bin",
                       "frames_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world}, no collective"},
            "frames_per_s": round(fps, 2),
            "ms_per_frame_per_gpu": round(elapsed / args.steps / B * 1e3, 4),
            "mfma_frac_whole_path": round(fps / world * FLOP_PER_FRAME / PEAK_FP16_DENSE, 4),
            "mfma_frac_whole_path_direct_form": round(fps / world * FLOP_PER_FRAME_DIRECT / PEAK_FP16_DENSE, 4),
            "frames_per_launch": frames_per_launch,
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": PEAK_FP16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_FP16_DENSE, 4), "traffic": traffic,
                         "achieved_direct_form": round(achieved_direct, 2),
                         "frac_direct_form": round(achieved_direct * 1e12 / PEAK_FP16_DENSE, 4),
                         "note": "achieved = FLOPs of the kernel's algorithm (conv_up1 in its four-phase 3x3 form, "
                                 "2.78x fewer MACs than the reference's direct 5x5-on-upsampled form) x 1920x1080 x "
                                 "frames per launch / HIP-event launch time; *_direct_form counts conv_up1 as the "
                                 "reference evaluates it.  The chip holds about 1.8 GHz (not 2.4) under this MFMA "
                                 "load: a bare LDS-fed MFMA loop of the same shape measures 1.70 PFLOP/s here "
                                 "(tools/mfma_shape_bench.hip)"},
        }
        if shared:
            out["rehearsal"] = f"{world} ranks on {torch.cuda.device_count()} GPU(s): not a scaling measurement"
        info = ctx.device_info()
        out["device"] = {"arch": info["arch"], "compute_units": info["compute_units"], "max_clock_mhz": info["clock_khz"] // 1000,
                         "peak_note": "2.5 PFLOP/s = 256 CUs x 4096 fp16 FLOP/clk x 2.4 GHz"}
        if world == 1 and not args.no_latency_leg:
            out["latency_ms_single_frame"] = single_frame_latency(S, distinct[0], local_rank)
        if world == 1 and not args.no_pcie_leg:
            out["pcie_inclusive"] = pcie_inclusive(S, distinct, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
