#!/bin/bash
# end-to-end training bench: front end of the next batch on a side stream (default) against the serial form
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
timeout -k 10 300 python tools/bench_train.py 2>/dev/null | cut -c80-330 || exit 1
timeout -k 10 300 python tools/bench_train.py --serial 2>/dev/null | cut -c80-330 || exit 1
done
timeout -k 10 300 python tools/bench_train.py --batch 48 2>/dev/null | cut -c80-330 || exit 1
timeout -k 10 300 python tools/bench_train.py --batch 48 --serial 2>/dev/null | cut -c80-330
