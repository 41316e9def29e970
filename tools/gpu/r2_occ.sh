#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
SMH_FEAT_OCC=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep "occupancy" | head -2
