"""bench.py as the driver calls it for N > 1 without a launcher of its own: `python bench.py --gpus N`
must start the N ranks itself, before it touches torch or a GPU, relay rank 0's one JSON line and fail
when a rank fails.  Exercised here on the CPU with --dry-run (gloo rendezvous, the barrier and
max-over-ranks protocol of the real run, a stand-in ctx that launches nothing and reports no number)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          env=e, timeout=300)


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["value"] is None and "rehearsal" in out           # a dry run must never look like a measurement
    assert len(out["per_rank_seconds"]) == 2 and all(s > 0 for s in out["per_rank_seconds"])
    assert out["ms_per_step"] * 3 / 1e3 >= max(out["per_rank_seconds"]) * 0.98      # the slowest rank's time
    assert out["scaling"] == "weak" and out["config"]["frames_per_gpu_per_step"] == 64


def test_tiles_config_deals_eight_tiles():
    r = run(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--config", "tiles8k"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["tiles_per_gpu"] == 4 and out["config"]["halo"] == 7


def test_a_failing_rank_fails_the_run():
    r = run(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"], env={"SRX_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_under_a_launcher_still_forms_a_process_group():
    """RANK / WORLD_SIZE=1 / MASTER_PORT in the environment (what torch.distributed.run gives the only rank of
    `--nproc-per-node 1`): the barrier, the max-reduce and the all-gather run over a process group, as with 8."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = run(["--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0"],
            env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["timing_collectives"] == "gloo" and out["n_gpus"] == 1 and len(out["per_rank_seconds"]) == 1
    # without a launcher there is no group
    r = run(["--gpus", "1", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["timing_collectives"] is None
