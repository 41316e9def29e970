#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 400 python tools/time_model_sizes.py 2>&1 | grep -v amdgpu.ids
TIME_W=99 TIME_N=48,256,510,1024 timeout -k 10 400 python tools/time_model_sizes.py 2>&1 | grep -v amdgpu.ids
TIME_W=249 TIME_N=48,256,510 timeout -k 10 400 python tools/time_model_sizes.py 2>&1 | grep -v amdgpu.ids
