#!/bin/bash
# development helper: time the LR kernel under experiment variants (timing only; outputs are wrong for most)
for e in ${LR_VARIANTS:-0 1 2 4 8 16 32 64 128 192 208 12 14 6 10}; do
  echo "== EXP_LR=$e"
  SRX_EXP_LR=$e timeout -k 10 100 python tools/time_stages.py | grep pack_conv1 || exit 1
done
