#!/bin/bash
# round 4: the ragged front end as one launch per stage -- timings of four file-length mixes and the per-stage kernel times (rocprofv3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
: > $O/ragged_times.log
for args in "256 10" "256 3" "64 30" "32 120"; do
  timeout -k 10 200 python3 tools/time_ragged.py $args >> $O/ragged_times.log 2>&1 || exit 1
done
rm -rf gpurun_out/prof/rag && mkdir -p gpurun_out/prof/rag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/rag -- python3 tools/time_ragged.py 256 10 > $O/ragged_prof.log 2>&1
echo "rocprof rc=$?"
f=$(ls gpurun_out/prof/rag/*/*kernel_stats.csv | head -1)
cp "$f" $O/ragged_kernel_stats.csv
cat $O/ragged_times.log
head -20 $O/ragged_kernel_stats.csv
