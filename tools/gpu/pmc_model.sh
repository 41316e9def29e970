#!/bin/bash
# matrix-core / stall counters of the network kernel inside a short bench run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
PAT=${1:-b3mtl_forward}
rm -rf gpurun_out/prof/m && mkdir -p gpurun_out/prof/m
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/m/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/m/$n.log 2>&1; echo "$n rc=$?"; tail -2 gpurun_out/prof/m/$n.log | cut -c1-200; }
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES
pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
pass p3 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
python3 - "$PAT" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
for d in ("p1", "p2", "p3"):
    fs = glob.glob("gpurun_out/prof/m/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("%-28s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
exit 0
