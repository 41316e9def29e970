#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/tune_split.log; : > $L
run() { echo "== $*" >> $L; env "$@" timeout -k 10 120 python tools/median_only.py >> $L 2>&1 || exit 1; }
for pair in "17 17" "21 11"; do
  set -- $pair
  run LH=$1 LP=$2 SMH_MEDIAN_NOSPLIT=1
  run LH=$1 LP=$2 SMH_MEDIAN_PERSIST=0
  run LH=$1 LP=$2 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=512
  run LH=$1 LP=$2 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=768
  run LH=$1 LP=$2 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=768 SMH_MEDIAN_SEG=2,2
  run LH=$1 LP=$2 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=768 SMH_MEDIAN_SEG=1,3
done
run LH=17 LP=17 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=1024
run LH=17 LP=17 SMH_MEDIAN_PERSIST=1 SMH_MEDIAN_PTHREADS=1024 SMH_MEDIAN_SEG=2,3
grep -v amdgpu.ids $L
