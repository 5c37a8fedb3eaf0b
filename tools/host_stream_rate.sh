#!/bin/bash
# On the GPU box: the C++ host's frame stream (srx_main -stream) at 1080p, next to bench.py's host-to-host leg.
#   bash tools/host_stream_rate.sh [frames]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-1920}
W=$(mktemp -d)
ln -s "$R/superresolution_amd/Assets" "$W/Assets"
python3 - "$W/hd.png" <<'PY'
import sys
import numpy as np
from PIL import Image
rng = np.random.default_rng(1080)
noise = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.int64)
yy, xx = np.mgrid[0:1080, 0:1920]
Image.fromarray(((noise + ((xx + yy) & 255)[..., None]) // 2).astype(np.uint8)).save(sys.argv[1], compress_level=1)
PY
cd "$W" || exit 1
for k in 1 2 3; do
    timeout -k 10 120 "$R/superresolution_amd/srx_main" -i hd.png -stream "$N" -noshot | grep "stream on device" || exit 1
done
