#!/bin/bash
# whole -m gpu suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2/all_gpu_tests.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/all_gpu_tests.log | tail -8 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
