#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python tools/time_train.py 2>&1 | grep -v amdgpu.ids | tail -12
