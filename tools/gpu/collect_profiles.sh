#!/bin/bash
# Round profile set: kernel-trace stats of the default bench + PMC traffic passes (FETCH_SIZE, WRITE_SIZE separately).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${1:-r03}
rm -rf gpurun_out/prof/final && mkdir -p gpurun_out/prof/final
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/final/trace3 -- python3 bench.py --workload config3 > gpurun_out/prof/final/bench_trace3.log 2>&1; echo "trace3 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/final/trace -- python3 bench.py --no-cpu-baseline > gpurun_out/prof/final/bench_trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/final/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > gpurun_out/prof/final/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/final/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > gpurun_out/prof/final/pmc_write.log 2>&1; echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/prof/final/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > gpurun_out/prof/final/pmc_sq.log 2>&1; echo "sq rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof/final/pmc_sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > gpurun_out/prof/final/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
python3 - "$R" <<'PY'
import csv, glob, json, collections, sys
R = sys.argv[1]
short = {"stft": "stft", "hpss_median": "median", "preprocess_fused_kernel": "preprocess_signal", "hp_feat": "hp_feat", "features_clip_kernel": "features_clip", "features_half_kernel": "features_half", "std_patch_kernel": "std_patch", "b3mtl_forward_kernel": "model"}
def key(name):
    for k, v in short.items():
        if k in name: return v
    return None
out = collections.defaultdict(dict)
for d, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", None), ("pmc_sq2", None)):
    fs = glob.glob("gpurun_out/prof/final/%s/*/*_counter_collection.csv" % d)
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        k = key(r["Kernel_Name"])
        if k: agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        out[k][c] = sum(v) / len(v)
for k, c in out.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM)
        c["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
fs = glob.glob("gpurun_out/prof/final/trace/*/*_kernel_stats.csv")
if fs:
    for r in csv.DictReader(open(fs[0])):
        k = key(r["Name"])
        if k: out[k]["avg_ns_rocprof"] = float(r["AverageNs"]); out[k]["calls"] = int(r["Calls"])
json.dump(out, open("gpurun_out/prof/final/%s_pmc_summary.json" % R, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
cp $(ls gpurun_out/prof/final/trace/*/*_kernel_stats.csv | head -1) gpurun_out/prof/final/${R}_kernel_stats.csv
cp $(ls gpurun_out/prof/final/trace3/*/*_kernel_stats.csv | head -1) gpurun_out/prof/final/${R}_kernel_stats_config3.csv
grep "^{\"metric\"" gpurun_out/prof/final/bench_trace3.log | tail -1 > gpurun_out/prof/final/${R}_bench_config3_under_rocprof.json
grep "^{\"metric\"" gpurun_out/prof/final/bench_trace.log | tail -1 > gpurun_out/prof/final/${R}_bench_under_rocprof.json
exit 0
