#!/bin/bash
# One rocprofv3 counter pass over tools/time_stages.py (a pass of eight 1080p frames), on the GPU box.
#   bash tools/pmc_pass.sh <name> <counter> [<counter> ...]       (environment such as SRX_LIB / SRX_EXP is inherited)
# Prints the median per kernel and counter; raw CSV under gpurun_out/pmc_<name>/.
NAME=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT" -o pmc --output-format csv -- \
    python3 "$R/tools/time_stages.py" --batch 8 --reps 3 > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, statistics, sys
vals = {}
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        for k in ("LrFusedKernel", "HrFusedKernel"):
            if k in row["Kernel_Name"]:
                vals.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    print(f"{k:14s} {c:28s} {statistics.median(v):16.1f}")
PY
