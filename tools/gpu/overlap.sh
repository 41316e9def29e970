#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python tools/overlap_probe.py > gpurun_out/overlap.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/overlap.log | tail -8
exit $rc
