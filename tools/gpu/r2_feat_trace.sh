#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
TRACE_OUT=gpurun_out/r2/feat_trace.npy timeout -k 10 300 python tools/trace_features.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/feat_trace.txt
