#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 300 python tools/time_features.py 2>&1 | grep -v amdgpu.ids | grep "probe" | tee gpurun_out/r2/time_features_probe.log
