#!/bin/bash
# On the GPU box: which device this is, what the benchmark measures on it, and the clocks the two fused kernels hold
# (diagnostic library, in-kernel clock probe).  One line of JSON on stdout; run it on several boxes to see the spread.
#   bash tools/device_survey.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
ID=$(rocm-smi --showuniqueid 2>/dev/null | grep -o "0x[0-9a-f]*" | head -1)
timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pcie-leg --no-latency-leg > /tmp/survey_bench.json 2>/dev/null || exit 1
SRX_LIB=$R/superresolution_amd/libsrx_diag.so SRX_CLOCK=1 timeout -k 10 120 python3 tools/time_stages.py --batch 8 --reps 45 > /tmp/survey_stages.txt 2> /tmp/survey_clock.txt || exit 1
python3 - "$ID" <<'PY'
import json, re, sys
b = json.load(open("/tmp/survey_bench.json"))
clk = {}
for line in open("/tmp/survey_clock.txt"):
    m = re.search(r"(LrFusedKernel|HrFusedKernel).*?([0-9.]+) GHz", line)
    if m and m.group(1) not in clk:
        clk[m.group(1)] = line.strip()[:260]
print(json.dumps({"device_unique_id": sys.argv[1], "frames_per_s": b["frames_per_s"], "stage_ms": b["stage_ms"],
                  "frac_issued_dominant": b["roofline"]["frac_issued"], "clock_probe": clk}))
PY
