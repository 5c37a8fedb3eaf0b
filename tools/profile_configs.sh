#!/bin/bash
# Runs on the GPU box: kernel-trace statistics and HBM-traffic counter passes (one counter per run) for the
# two configurations besides the frame stream: one band of the 7680x4320 image as the tile partition runs it
# (7680 x 554 rows: 540 + a 7-row halo on either side, batch 1) and the single 1920x1080 frame (batch 1).
#   bash tools/profile_configs.sh r03
# Output under gpurun_out/prof_<tag>_<config>/; tools/summarize_profiles.py <tag> --configs folds it into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run_config() {
    NAME=$1; shift
    OUT=$R/gpurun_out/prof_${TAG}_$NAME
    mkdir -p "$OUT"
    timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- \
        python3 "$R/tools/time_stages.py" "$@" --reps 20 > "$OUT/stages.txt" 2> "$OUT/stats.log" || return 1
    for C in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 120 rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc --output-format csv -- \
            python3 "$R/tools/time_stages.py" "$@" --reps 3 > "$OUT/pmc_$C.log" 2>&1 || return 1
    done
}
run_config tiles8k --w 7680 --h 554 --batch 1 && run_config latency --w 1920 --h 1080 --batch 1 && echo done
