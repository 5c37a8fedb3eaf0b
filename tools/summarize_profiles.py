#!/usr/bin/env python3
"""Turns what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the summaries kept in
profiles/: kernel-trace statistics, the bench line printed under rocprofv3, the median counter
values per kernel, and the HBM traffic per launch that bench.py reports as roofline.traffic.

  python tools/summarize_profiles.py r01 [prefix]      # prefix of the files in profiles/, default = tag
  python tools/summarize_profiles.py r03 --configs     # what tools/profile_configs.sh left: the tile and the
                                                       # single-frame configurations, added to profiles/traffic.json
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


CONFIG_NOTES = {
    "tiles8k": "one band of the 7680x4320 image as the tile partition runs it: 7680 x 554 input rows (540 + a 7-row halo on "
               "either side), batch 1, one launch of each fused kernel",
    "latency": "one 1920x1080 frame, batch 1 (strips cut into row segments), one launch of each fused kernel",
}


def configs(tag):
    """Kernel statistics and HBM bytes per launch of the configurations profiled by tools/profile_configs.sh."""
    dst = os.path.join(ROOT, "profiles")
    tp = os.path.join(dst, "traffic.json")
    traffic = json.load(open(tp))
    for name, note in CONFIG_NOTES.items():
        src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{name}")
        shutil.copy(os.path.join(src, "stats", "run_kernel_stats.csv"), os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "stages.txt"), os.path.join(dst, f"{tag}_{name}_stage_times.txt"))
        raw = {k: {} for k in KERNELS}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            values = {}
            for row in csv.DictReader(open(os.path.join(src, "pmc_" + counter, "pmc_counter_collection.csv"))):
                for k, kernel in KERNELS.items():
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        values.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in values.items():
                raw[k][counter] = statistics.median(v)
        entry = {"_config": note, "_measured": f"profile set {tag}, same corrections as the stream entries", "_raw_KiB": raw}
        for k in KERNELS:
            entry[STAGE[k]] = int(round((2 * raw[k]["FETCH_SIZE"] + raw[k]["WRITE_SIZE"]) * 1024))
        traffic[name] = entry
    json.dump(traffic, open(tp, "w"), indent=1)
    print(json.dumps({k: traffic[k] for k in CONFIG_NOTES}, indent=1))


def main():
    tag = sys.argv[1]
    if "--configs" in sys.argv:
        return configs(tag)
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
    tp = os.path.join(dst, "traffic.json")
    if os.path.exists(tp):                  # keep the entries of the other configurations
        old = json.load(open(tp))
        traffic.update({k: old[k] for k in CONFIG_NOTES if k in old})
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
