#!/bin/bash
# walk kernel: files of even / odd / mixed frame counts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for par in 0 1 ""; do
  rm -rf gpurun_out/prof/ragp && mkdir -p gpurun_out/prof/ragp
  RAG_T_PARITY=$par timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ragp -- python3 tools/time_ragged.py 256 10 > gpurun_out/r4/ragp_$par.log 2>&1
  echo "parity=$par"; grep "^ragged" gpurun_out/r4/ragp_$par.log
  f=$(ls gpurun_out/prof/ragp/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("rag_", "median_split", "stft400", "features_")) and "true" in n or "rag_" in n or "median_split" in n or "stft400" in n:
        print("  %-40s calls %3s avg %8.1f us" % (n.split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
