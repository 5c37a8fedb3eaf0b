#!/bin/bash
# development helper: time the LR kernel under experiment variants (timing only; outputs are wrong for most)
for e in ${LR_VARIANTS:-0 2 4 8 12 6}; do
  echo "== EXP_LR=$e"
  SRX_CLOCK=1 SRX_EXP_LR=$e timeout -k 10 100 python tools/time_stages.py 2>&1 | grep -E "pack_conv1|Lr.*clock" || exit 1
done
