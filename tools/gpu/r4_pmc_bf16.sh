#!/bin/bash
# SQ counters of the split-bf16 network kernel (tools/time_bf16_blocks.py: 1024 patches from layer-0 partials, 5-class)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/bf && mkdir -p gpurun_out/prof/bf
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/bf/$n -- python3 tools/time_bf16_blocks.py > gpurun_out/prof/bf/$n.log 2>&1; echo "$n rc=$?"; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
pass p3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    fs = glob.glob("gpurun_out/prof/bf/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "bf16s" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("%-32s %.5g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
