"""The RCCL branch of the multi-GPU path, executed on the one GPU of the test box: a process group with the
nccl backend and world size 1 goes through the same torch.distributed calls as the eight-rank run (init with
a device id, barrier, max-reduce, all-gather, the optional gather of tile interiors on device tensors).
Every case runs in a fresh child process: a process group is process-wide state, and a process that has
initialised the GPU is never re-executed."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_env():
    e = dict(os.environ)
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return e


def _run_rank(cmd):
    """Run `cmd` as the only rank of a job.  The rendezvous port is picked by binding port 0 and closing it, so now and
    then another process takes it first: that one failure (EADDRINUSE, before any GPU work) is retried on a new port."""
    for _ in range(3):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_rank_env(), timeout=300)
        if r.returncode == 0 or "EADDRINUSE" not in r.stderr:
            break
    return r


def test_sharding_helpers_over_nccl_world_size_1():
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np
        import torch
        import torch.distributed as dist
        from superresolution_amd import sharding as sh
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        dist.barrier()
        torch.cuda.synchronize()
        assert sh.max_over_ranks(1.25) == 1.25
        assert sh.gather_over_ranks(7.5) == [7.5]
        W, H = 50, 38
        truth = np.random.default_rng(5).integers(0, 256, (2 * H, 2 * W, 4), dtype=np.uint8)
        plan = sh.tile_plan(W, H, 3, 2)
        interiors = [np.ascontiguousarray(truth[2 * t.y0:2 * t.y1, 2 * t.x0:2 * t.x1]) for t in plan]
        whole = sh.gather_tiles(interiors, W, H, 3, 2, 0, 1, dst=0, device="cuda")
        assert np.array_equal(whole, truth)
        # and with real tiles: the halo tiles of an image through the GPU path, gathered on the device
        import superresolution_amd as S
        from superresolution_amd.sharding import upscale_tiled
        rng = np.random.default_rng(9)
        img = rng.integers(0, 256, (40, 66, 4), dtype=np.uint8)
        img[..., 3] = 255
        with S.SuperResolutionContext(66, 40) as ctx:
            full = ctx.run(img)
        tiles = sh.tile_plan(66, 40, 2, 2)
        outs = []
        for t in tiles:
            with S.SuperResolutionContext(t.ex1 - t.ex0, t.ey1 - t.ey0) as ctx:
                outs.append(np.ascontiguousarray(t.crop(ctx.run(np.ascontiguousarray(img[t.ey0:t.ey1, t.ex0:t.ex1])))))
        assert np.array_equal(sh.gather_tiles(outs, 66, 40, 2, 2, 0, 1, dst=0, device="cuda"), full)
        dist.barrier()
        dist.destroy_process_group()
        print("rccl-ok")
    """)
    r = _run_rank([sys.executable, "-c", code])
    assert r.returncode == 0 and "rccl-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_rank_under_a_launcher_uses_nccl():
    """bench.py as rank 0 of a one-rank job (the environment torch.distributed.run gives a rank): process group
    over nccl, barrier, max over ranks and the all-gather of the per-rank times, then the usual line."""
    r = _run_rank([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                   "--batch", "8", "--no-cpu-baseline", "--no-pcie-leg", "--no-latency-leg", "--stage-reps", "2"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["timing_collectives"] == "nccl" and out["n_gpus"] == 1
    assert out["value"] > 0 and len(out["per_rank_seconds"]) == 1 and len(out["per_rank_frames_per_s"]) == 1
