#!/bin/bash
# SQ / memory PMC passes for the median kernel alone (tools/median_only.py); extra env is passed through.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/med && mkdir -p gpurun_out/prof/med
timeout -k 10 200 python tools/median_only.py > gpurun_out/prof/med/plain.log 2>&1; echo "plain rc=$?"; tail -1 gpurun_out/prof/med/plain.log
rocprofv3 --list-avail > gpurun_out/prof/med/avail.txt 2>&1
grep -o -i -E "\b(SQ_IFETCH[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQ_[A-Z_]*STALL[A-Z_]*|TCP_[A-Z_]*STALL[A-Z_]*|TA_BUSY[A-Z_]*|TCC_[A-Z_]*STALL[A-Z_]*|TCP_TCC_WRITE[A-Z_]*|TCC_EA0_WRREQ[A-Z0-9_]*|SQ_INSTS_SMEM|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_SCA|SQ_INST_CYCLES_SALU|SQ_THREAD_CYCLES_VALU)\b" gpurun_out/prof/med/avail.txt | sort -u | tr '\n' ' '; echo
export ITERS=3
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/med/$n -- python3 tools/median_only.py > gpurun_out/prof/med/$n.log 2>&1; echo "$n rc=$?"; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
pass p3 FETCH_SIZE
pass p4 WRITE_SIZE
pass p5 SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
pass p6 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2", "p3", "p4", "p5", "p6"):
    fs = glob.glob("gpurun_out/prof/med/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "median" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("%-28s %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
exit 0
