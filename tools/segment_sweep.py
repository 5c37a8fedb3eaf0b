#!/usr/bin/env python3
"""How many row segments should a lone 1080p frame be cut into?  (development helper, GPU box)

Runs tools/time_stages.py --batch 1 once per candidate in a child process with the diagnostic build of the
library (make -C superresolution_amd/csrc DIAG=1), whose SRX_SEGS_LR / SRX_SEGS_HR force the number of row
segments of the two fused kernels, and prints the time of each kernel's launch per candidate.

  python tools/segment_sweep.py [--lr 4,6,7,8,9,10,12,16] [--hr 4,6,8,10,12,16]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stage_times(env):
    e = dict(os.environ, SRX_LIB=os.path.join(ROOT, "superresolution_amd", "libsrx_diag.so"), **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_stages.py"), "--batch", "1", "--reps", "50"],
                         env=e, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    t = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(\S+)\s+([0-9.]+) ms/frame", line)
        if m:
            t[m.group(1)] = float(m.group(2))
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", default="4,6,7,8,9,10,12,16")
    ap.add_argument("--hr", default="4,6,8,10,12,16")
    a = ap.parse_args()
    base = stage_times({})
    print("planner's choice:", base, flush=True)
    for n in [int(x) for x in a.lr.split(",")]:
        print(f"LR segments {n:2d}: {stage_times({'SRX_SEGS_LR': str(n)})['pack_conv1_conv2_conv3_conv_up1']:.4f} ms", flush=True)
    for n in [int(x) for x in a.hr.split(",")]:
        print(f"HR segments {n:2d}: {stage_times({'SRX_SEGS_HR': str(n)})['conv4_conv5_conv6_add_unorm']:.4f} ms", flush=True)


if __name__ == "__main__":
    main()
