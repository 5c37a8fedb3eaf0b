#!/bin/bash
for m in ${PRIOS:-0 1 8 14 12 9}; do
  echo "== PRIO=$m"
  SRX_CLOCK=1 SRX_LR_PRIO=$m timeout -k 10 100 python tools/time_stages.py --batch 8 2>&1 | grep -E "pack_conv1|Lr.*clock" || exit 1
done
