#!/bin/bash
# Runs on the GPU box: kernel-trace stats of bench.py and the HBM-traffic counter passes
# (one counter group per run), written under gpurun_out/ for copying into profiles/.
#   bash tools/profile_round.sh r01
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-leg --no-latency-leg --stage-reps 10 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log" || exit 1
for C in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    timeout -s KILL 120 rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc --output-format csv -- \
        python3 "$R/tools/time_stages.py" --batch 8 --reps 3 > "$OUT/pmc_$C.log" 2>&1 || exit 1
done
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    -d "$OUT/pmc_SQ" -o pmc --output-format csv -- python3 "$R/tools/time_stages.py" --batch 8 --reps 3 > "$OUT/pmc_SQ.log" 2>&1 || exit 1
echo done
