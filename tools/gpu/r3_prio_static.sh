#!/bin/bash
# round 3: static s_setprio for waves 4..7 (the younger wave of every SIMD), probe SMH_TCN_TUNE bits 17..18
cd "$GRAFT_REPO_ROOT" || exit 1
export SMH_ENABLE_PROBES=1
for p in 0 1 2 3 0 1 3; do SMH_TCN_TUNE=$((p << 17)) timeout -k 10 120 python3 tools/model_only.py 1024 300 2>/dev/null || exit 1; done
for p in 0 3; do echo "== stamps prio $p"; SMH_TCN_TUNE=$(((p << 17) | 128)) timeout -k 10 200 python3 tools/trace_model.py 2>/dev/null | grep -E "fetched ahead|wave [0-7]:" | head -9; done
