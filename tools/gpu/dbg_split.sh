#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/dbg_split.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 120 python tools/median_only.py >> $L 2>&1 || exit 1; }
run LH=17 LP=17 SMH_DBG=0
run LH=17 LP=17 SMH_DBG=1
run LH=17 LP=17 SMH_DBG=2
run LH=17 LP=17 SMH_DBG=1 SMH_MEDIAN_PTHREADS=768
run LH=17 LP=17 SMH_DBG=2 SMH_MEDIAN_PTHREADS=768
grep -v amdgpu.ids $L
