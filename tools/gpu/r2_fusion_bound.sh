#!/bin/bash
# upper bound of an STFT -> median fusion: the two kernels with S's store / S's load taken out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
{
for rep in 1 2; do
timeout -k 10 200 python tools/fusion_bound.py 2>&1 | grep probes || exit 1
SMH_STFT_PROBE_NOSTORE=1 timeout -k 10 200 python tools/fusion_bound.py 2>&1 | grep probes || exit 1
SMH_MEDIAN_PROBE_NOLOAD=1 timeout -k 10 200 python tools/fusion_bound.py 2>&1 | grep probes || exit 1
SMH_STFT_PROBE_NOSTORE=1 SMH_MEDIAN_PROBE_NOLOAD=1 timeout -k 10 200 python tools/fusion_bound.py 2>&1 | grep probes || exit 1
done
} | tee gpurun_out/r2/fusion_bound.txt
