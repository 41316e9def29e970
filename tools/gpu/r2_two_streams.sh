#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python tools/two_stream_probe.py 2>&1 | grep -v amdgpu.ids | tail -5
