"""A MODEL of the 1 / 2 / 4 / 8-GPU throughput of both readings of BASELINE config 4 and of the inference bench -- NOT a
measurement: the builder's box has one GPU, and the driver's 8-GPU node is the only place a curve can be measured.

What is measured here, on ONE MI355X (child processes; this parent never touches the GPU):
  * the inference step of `bench.py` at B = 1024 clips (the headline path: no data-path collective);
  * the end-to-end training step of `tools/bench_train.py` (front end -> patches -> forward, losses, backward, optimiser) at the
    local batch a rank would hold: 510 clips (config 4 read as "512 per GPU", weak scaling) and 510 / G for G = 2, 4, 8
    (config 4 read literally: 512 GLOBAL over 8 GPUs, strong scaling), batches rounded to the multiple of 3 the class-balanced
    batch needs;
  * the RCCL all-reduce of the real gradient bucket (218 839 floats = 0.875 MB at W = 68; 514 231 floats = 2.06 MB at W = 249)
    on a ONE-rank process group: RCCL's launch + kernel floor, without any link.
What is ASSUMED (stated in the output): the xGMI part of the all-reduce, t_link(G) = 2 (G - 1) alpha + 2 (G - 1) / G * bytes / beta,
between an optimistic (alpha 1.5 us per ring step, beta = 7 links x 153 GB/s x 0.7) and a pessimistic (alpha 5 us, one link at
100 GB/s) corner; that the ranks do not share a host resource (one process per GPU, 16 cores each on the test pool); and that
the all-reduce is exposed (it sits on the main stream between the backward pass and the optimiser; only the NEXT batch's front
end overlaps it).

    python tools/scaling_model.py [--steps 200] [--dtype bf16] > profiles/r04_scaling_model.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ALLREDUCE = r'''
import os, json, sys, time, socket
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
out = {}
for name, n in (("w68", 218743 + 96), ("w249", 514135 + 96)):
    t = torch.randn(n, device="cuda")
    for _ in range(50):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        dist.all_reduce(t)
    e1.record()
    torch.cuda.synchronize()
    out[name] = {"floats": n, "bytes": 4 * n, "us_per_call_one_rank": e0.elapsed_time(e1) / 500 * 1e3}
dist.destroy_process_group()
print(json.dumps(out))
'''


def run_json(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stderr[-2000:]))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="the training legs' matrix pipe (tools/bench_train.py --dtype)")
    args = ap.parse_args()
    py = sys.executable
    sw = ["--steps", str(args.steps), "--warmup", str(args.warmup)]
    inf = run_json([py, "bench.py", "--no-cpu-baseline", "--steady-steps", "0", *sw])
    train = {}
    for b in (510, 255, 126, 63):
        j = run_json([py, "tools/bench_train.py", "--batch", str(b), "--serial", "--dtype", args.dtype, *sw])
        train[b] = {"ms_per_step": j["ms_per_step"], "clips_per_step": j.get("config", {}).get("clips_per_gpu_per_step", b)}
    ar = run_json([py, "-c", ALLREDUCE])

    def t_link(G, nbytes, alpha_us, beta_gbs):
        return 0.0 if G == 1 else 2 * (G - 1) * alpha_us + 2.0 * (G - 1) / G * nbytes / (beta_gbs * 1e3)  # us

    corners = {"optimistic": (1.5, 7 * 153 * 0.7), "pessimistic": (5.0, 100.0)}
    nbytes = ar["w68"]["bytes"]
    floor = ar["w68"]["us_per_call_one_rank"]
    model = {"inference_weak_B1024_per_gpu": {}, "training_weak_510_per_gpu": {}, "training_strong_510_global": {}}
    for G in (1, 2, 4, 8):
        model["inference_weak_B1024_per_gpu"][str(G)] = {"clips_per_s": G * 1024 / (inf["ms_per_step"] * 1e-3), "efficiency": 1.0,
                                                         "note": "no collective inside the timed steps"}
        local = {1: 510, 2: 255, 4: 126, 8: 63}[G]
        for key, b in (("training_weak_510_per_gpu", 510), ("training_strong_510_global", local)):
            row = {"local_clips": b, "ms_compute": train[b]["ms_per_step"]}
            for cname, (a_us, beta) in corners.items():
                t_ar = 0.0 if G == 1 else floor + t_link(G, nbytes, a_us, beta)
                ms = train[b]["ms_per_step"] + t_ar * 1e-3
                row[cname] = {"allreduce_us": round(t_ar, 1), "ms_per_step": round(ms, 4), "clips_per_s": round(G * b / (ms * 1e-3))}
            one = model[key].get("1")
            if one:
                for cname in corners:
                    row[cname]["speedup_vs_1"] = round(row[cname]["clips_per_s"] / one[cname]["clips_per_s"], 2)
            model[key][str(G)] = row
    print(json.dumps({
        "what": "MODEL, NOT A MEASUREMENT: 1/2/4/8-GPU throughput predicted from one-GPU step times + the one-rank RCCL all-reduce floor "
                "+ an assumed xGMI term (see tools/scaling_model.py)",
        "training_dtype": args.dtype,
        "measured_on_one_gpu": {"inference_ms_per_1024_clips": inf["ms_per_step"], "inference_clips_per_s": inf["value"],
                                "training_step_ms_by_local_batch": train, "rccl_allreduce_one_rank": ar},
        "assumptions": {"t_link_us": "2 (G-1) alpha + 2 (G-1)/G * bytes / beta", "corners_alpha_us_beta_GBs": corners,
                        "bucket_bytes": nbytes, "allreduce_exposed": True, "hosts_independent": True},
        "model": model}, indent=1))


if __name__ == "__main__":
    main()
