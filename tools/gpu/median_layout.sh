#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/median_layout.log; : > $L
for lay in 1 2; do for pair in "17 17" "21 11"; do set -- $pair; LAYOUT=$lay LH=$1 LP=$2 timeout -k 10 120 python tools/median_only.py >> $L 2>&1 || exit 1; done; done
grep -v amdgpu.ids $L
