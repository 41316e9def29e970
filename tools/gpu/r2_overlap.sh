#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
for v in "9 0" "8 0" "8 1"; do
  set -- $v
  echo "== waves=$1 prefetch=$2"
  SMH_TCN_WAVES=$1 SMH_TCN_PREFETCH=$2 timeout -k 10 200 python tools/overlap_stft_model.py 2>&1 | grep -v amdgpu.ids | tail -4
done
echo "== waves=8 prefetch=0, 8-frame stft workgroups"
SMH_STFT_FRAMES=8,256 SMH_TCN_WAVES=8 SMH_TCN_PREFETCH=0 timeout -k 10 200 python tools/overlap_stft_model.py 2>&1 | grep -v amdgpu.ids | tail -4
