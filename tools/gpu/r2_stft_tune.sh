#!/bin/bash
# STFT: frames per workgroup / threads per workgroup sweep (SMH_STFT_FRAMES=maxf,nthreads), stage time from the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('stft','median')})"; }
for v in 25,256 20,256 17,256 16,256 14,256 25,256; do
SMH_STFT_FRAMES=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | show "frames,threads=$v" || exit 1
done
