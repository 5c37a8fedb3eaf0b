#!/usr/bin/env python3
"""Turns what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the summaries kept in
profiles/: kernel-trace statistics, the bench line printed under rocprofv3, the median counter
values per kernel, and the HBM traffic per launch that bench.py reports as roofline.traffic.

  python tools/summarize_profiles.py r01 [prefix]      # prefix of the files in profiles/, default = tag
"""
import csv
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"LR": "LrFusedKernel", "HR": "HrFusedKernel"}
STAGE = {"LR": "pack_conv1_conv2_conv3_conv_up1", "HR": "conv4_conv5_conv6_add_unorm"}


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else tag
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, prefix + "_fused_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, prefix + "_bench_under_rocprof.json"))
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, prefix + "_bench.json"))
    summary = {k: {} for k in KERNELS}
    for d in sorted(os.listdir(src)):
        path = os.path.join(src, d, "pmc_counter_collection.csv")
        if not d.startswith("pmc_") or not os.path.exists(path):
            continue
        values = {}
        for row in csv.DictReader(open(path)):
            for k, name in KERNELS.items():
                if name in row["Kernel_Name"]:
                    values.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
        for (k, counter), v in values.items():
            summary[k][counter] = statistics.median(v)      # per launch of eight 1080p frames
    json.dump(summary, open(os.path.join(dst, prefix + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    import datetime
    import subprocess
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    traffic = {"frames_per_launch": 8,
               "_measured": f"profile set {tag}, {datetime.date.today().isoformat()}, tree at or after commit {head}",
               "_note": "HBM bytes per launch (eight 1080p frames): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                        "separate passes over tools/time_stages.py --batch 8, KiB -> bytes, FETCH_SIZE doubled (gfx950 "
                        "tallies the 128-B requests of 16-B-per-lane loads at 64 B)",
               "_raw_KiB": {k: {c: summary[k][c] for c in ("FETCH_SIZE", "WRITE_SIZE")} for k in KERNELS}}
    for k in KERNELS:
        traffic[STAGE[k]] = int(round((2 * summary[k]["FETCH_SIZE"] + summary[k]["WRITE_SIZE"]) * 1024))
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
