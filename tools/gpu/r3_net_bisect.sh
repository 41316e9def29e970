#!/bin/bash
mkdir -p gpurun_out/r3
for v in 0 1; do
  echo "== SMH_TCN_SKEW16=$v"
  SMH_TCN_SKEW16=$v timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -q -k "b3mtl_forward_vs_oracle or schedules_agree" 2>&1 | tail -15
done
