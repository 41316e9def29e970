#!/bin/bash
# round 3: skew schedule, wave k starts k x d late (probe SMH_TCN_TUNE bits 13..16, d in 256-cycle units) -- launch time, then the stamps
cd "$GRAFT_REPO_ROOT" || exit 1
export SMH_ENABLE_PROBES=1
for k in 0; do
  SMH_TCN_TUNE=$((k << 13)) timeout -k 10 120 python3 tools/model_only.py 1024 300 2>/dev/null || exit 1
done
for k in 0 3; do
  echo "== stamps, stagger $k"
  SMH_TCN_TUNE=$(((k << 13) | 128)) timeout -k 10 200 python3 tools/trace_model.py 2>/dev/null | grep -E "fetched ahead|task total|blocks of dilation|loop cycles" | head -14
done
