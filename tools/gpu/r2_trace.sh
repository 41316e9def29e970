#!/bin/bash
# B3_MTL forward: in-kernel timeline of the block loop (tools/trace_model.py) for the skewed and the barrier schedule
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
for v in "1 8 0" "0 9 0" "1 4 0" "1 8 128" "1 4 128"; do
set -- $v
echo "== skew=$1 waves=$2 tune=$3"
SMH_TCN_SKEW=$1 SMH_TCN_WAVES=$2 SMH_TCN_TUNE=$3 TRACE_OUT=gpurun_out/r2/tcn_trace_$3.npy timeout -k 10 200 python tools/trace_model.py 2>&1 | grep -v amdgpu.ids | grep -v "mod 8\|RuntimeWarning\|_methods\|ret = \|print(" || exit 1
done
