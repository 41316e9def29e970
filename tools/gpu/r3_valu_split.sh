#!/bin/bash
# round 3: where the network kernel's VALU instructions are -- the whole kernel against the kernel with 0 residual blocks (probe)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/vs && mkdir -p gpurun_out/prof/vs
for nb in 24 0 8; do
  if [ $nb = 24 ]; then unset SMH_TCN_BLOCKS SMH_ENABLE_PROBES; else export SMH_ENABLE_PROBES=1 SMH_TCN_BLOCKS=$nb; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof/vs/b$nb -- python3 tools/model_only.py > gpurun_out/prof/vs/b$nb.log 2>&1; echo "blocks $nb rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for nb in (24, 8, 0):
    fs = glob.glob("gpurun_out/prof/vs/b%d/*/*_counter_collection.csv" % nb)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "b3mtl_forward" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("blocks %2d:" % nb, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(agg.items())}, "(millions per launch)")
PY
