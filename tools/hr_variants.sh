#!/bin/bash
# development helper: time the HR kernel under experiment variants (timing only; outputs are wrong for most)
for cu in 2 1; do
for e in ${HR_VARIANTS:-0 4 8 64 256 512 768 12 24 28 40 56 60}; do
  echo "== EXP=$e PER_CU=$cu"
  SRX_EXP=$e SRX_HR_PER_CU=$cu timeout -k 10 100 python tools/time_stages.py | grep conv4 || exit 1
done
done
