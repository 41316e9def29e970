#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic 1 s @ 16 kHz clips.

One "step" = one pass of the whole hot path over one batch that is already resident in HBM (`HotPath.step`,
sm_hpss_mtl_amd/pipeline.py -- the same object tests/test_bench_path_gpu.py puts under the oracle):
  STFT -> HPSS medians (l_harm x l_perc) -> soft masks -> mel -> dB -> standardise -> patches (W=68)
       -> B3_MTL forward (logits)
Workload = BASELINE.json configs[1] (batch 1024 x 1 s clips, 17x17 medians) carried through the
network forward, i.e. the metric "clips/sec HPSS+MTL-CNN fwd".

N > 1: one process per GPU.  `python bench.py --gpus N` starts its own N ranks (a child `torch.distributed.run`; this
parent never touches the GPU); under a launcher (RANK/WORLD_SIZE in the environment) the process is a rank.  Every
rank owns its own 1024 clips -- clips are independent units, so there is no collective on the data path (weak
scaling); only the timing barrier / MAX all-reduce use RCCL.  Asking for more GPUs than are visible is an error.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step; per-kernel figures for every
stage are in `kernels`: HIP events on the launch stream around every stage, recorded inside the timed region on every
10th step (`--event-every`; five markers per step cost ~10 us).  After the timed region the logits of each rank's first
clips are compared with tests/golden/bench_golden.npz (CPU-oracle logits of those very clips; data only) -- a bench
whose logits do not match the reference arithmetic fails instead of printing a number.  `cpu_baseline`
(rank 0, N = 1 only, before the GPU is touched) times the CPU path (numpy rfft + scipy.ndimage.median_filter + the
numpy restatement, `python -m oracle.cpu_baseline`) on one core and on all usable cores.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E spec peak (6.29 TB/s measured copy ceiling)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (= vector peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (only for --model-dtype bf16)
# algorithmic bytes / flops per clip, SURVEY 8(d) (K=201 bins, T=98 frames, 240 features, W=68)
K_BINS, T_FRAMES, FEAT, W_PATCH = 201, 98, 240, 68
BYTES = {
    "stft": 16000 * 4 + K_BINS * T_FRAMES * 4,                       # audio in, |S| out          142,792
    "median": 3 * K_BINS * T_FRAMES * 4,                             # S in, harm+perc out        236,376
    "features": 3 * K_BINS * T_FRAMES * 4 + 2 * FEAT * T_FRAMES * 4 + W_PATCH * FEAT * 4,  # S,h,p in; fv out, re-read; patches out
}
FLOPS_MODEL = 2.0 * (W_PATCH * FEAT * 32 + 24 * W_PATCH * (3 * 32 * 32 + 32 * 32) + W_PATCH * 32 * 51)  # 14.64 MFLOP
# full-chain tolerance of the golden check (oracle from audio: numpy f64 FFT; device: its own f32 STFT): the same
# abs 1e-4 SURVEY 8(d') asks of the network alone.  Measured on MI355X: 2.9e-6.
GOLDEN_LOGIT_TOL = 1e-4


def cpu_baseline(l_harm, l_perc, budget_s):
    """`python -m oracle.cpu_baseline` as a child process (numpy/scipy only, no GPU): 1 core + all usable cores."""
    r = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", "--l-harm", str(l_harm), "--l-perc", str(l_perc),
                        "--budget", str(budget_s)], cwd=ROOT, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed:\n" + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 + 200 steps = 0.1 s of GPU time.  A cold box runs its first ~20 steps 9 % slower (clock ramp: 20 timed steps
    # behind 3 warm-up steps read 0.410 ms, behind 50 or more 0.375 ms; 200 timed steps behind 3: 0.377 -- tools/gpu/r2_warm.sh)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--preroll", type=int, default=50,
                    help="untimed steps run BEFORE the W warm-up steps, so that a caller's short warm-up still meets a device at its "
                         "working clock; reported as preroll_steps")
    ap.add_argument("--batch", type=int, default=1024, help="clips per GPU per step")
    ap.add_argument("--l-harm", type=int, default=17)
    ap.add_argument("--l-perc", type=int, default=17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work per baseline leg")
    ap.add_argument("--no-fuse-l0", action="store_true",
                    help="write standardised patches and let the network read them (the reference's call structure) instead "
                         "of computing the network's first 1x1 convolution inside the feature kernel")
    ap.add_argument("--two-kernel-features", action="store_true",
                    help="time-major harm + hp_feat_walk / std_patch kernels instead of the single feature kernel")
    ap.add_argument("--event-every", type=int, default=10,
                    help="record the per-kernel HIP events on every n-th timed step (the markers of all five stages cost "
                         "about 10 us per step, 2 %% of it; the other timed steps run without them)")
    ap.add_argument("--classes", type=int, choices=[3, 5], default=3,
                    help="3: the headline B3_MTL (S, M, R, 3C); 5: the musan_5_class variant (S, M, N, R, 5C) of BASELINE config 5")
    ap.add_argument("--model-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = mixed-precision network (BASELINE config 5); NOT the parity path, never the default")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal: start the ranks, join the process group (SMH_DIST_BACKEND, e.g. gloo on a CPU "
                         "box), run the timing protocol around an empty step and report value = null.  Measures nothing.")
    return ap.parse_args(argv)


def dry_run(args, ranks):
    from sm_hpss_mtl_amd.launch import timed_region
    elapsed, ran = timed_region(ranks, lambda k, timed: time.sleep(0.001), args.steps, args.warmup, lambda: None)
    if ranks.rank == 0:
        print(json.dumps({"metric": "launcher dry run (no hot path executed)", "value": None, "unit": "clips/s",
                          "n_gpus": ranks.world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 4), "dry_run": True,
                          "backend": ranks.backend}))
    ranks.close()


def main():
    args = parse_args()
    from sm_hpss_mtl_amd.launch import init_ranks, spawn_ranks_if_needed, timed_region
    rc = spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:  # this process was the parent of the ranks: it never touched the GPU
        sys.exit(rc)

    # CPU baseline first (rank 0 of a one-GPU job): a child process, nothing of it overlaps the timed region
    cpu = None
    if args.gpus == 1 and not args.no_cpu_baseline and not args.dry_run and int(os.environ.get("RANK", "0")) == 0:
        cpu = cpu_baseline(args.l_harm, args.l_perc, args.cpu_budget)

    ranks = init_ranks(args.gpus)
    if args.dry_run:
        return dry_run(args, ranks)
    rank, world = ranks.rank, ranks.world

    import numpy as np
    import torch

    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.pipeline import STAGES, HotPath
    from sm_hpss_mtl_amd.sharding import shard_range
    from sm_hpss_mtl_amd.synth import synth_clips

    B = args.batch
    # weak scaling: the job is world*B clips; rank r owns the contiguous index range shard_range(...)
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    base = synth_clips(64, seed=1000 + rank)  # 64 distinct clips per rank, tiled to the batch
    audio = torch.from_numpy(np.tile(base, ((B + 63) // 64, 1))[:B]).cuda()
    dev = audio.device

    fe = Frontend(FrontendConfig(l_harm=args.l_harm, l_perc=args.l_perc))
    model = B3MTL(n_feat=FEAT, patch_size=W_PATCH, n_classes=args.classes, seed=0)
    hp = HotPath(fe, model, B, audio.shape[1], patch=W_PATCH, fuse_l0=not args.no_fuse_l0,
                 two_kernel_features=args.two_kernel_features, model_dtype=args.model_dtype)
    fuse_l0, want_lay = hp.fuse_l0, hp.want_layout
    names = list(STAGES)

    # HIP events on the launch stream (torch's current stream IS the stream every kernel is launched on)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    every = max(1, min(args.event_every, args.steps // 5))  # at least five samples when the caller asks for few steps
    sampled = [k for k in range(args.steps) if k % every == every - 1]
    sampled_set = set(sampled)

    def step(k, timed):
        hp.step(audio, ev[k] if (timed and k in sampled_set) else None)

    for _ in range(max(0, args.preroll)):
        hp.step(audio)
    elapsed, ran = timed_region(ranks, step, args.steps, args.warmup, torch.cuda.synchronize, dev)
    logits = hp.logits
    assert torch.isfinite(logits).all(), "non-finite logits"
    assert ran == world, "%d ranks ran the timed region, expected %d" % (ran, world)

    # ---- parity of the timed configuration: this rank's first clips against the committed oracle logits ----
    parity = {"checked": False}
    gpath = os.path.join(ROOT, "tests", "golden", "bench_golden.npz")
    gkey = "logits_%dx%d" % (args.l_harm, args.l_perc)
    if os.path.exists(gpath) and args.model_dtype == "f32" and args.classes == 3:
        g = np.load(gpath)
        n = int(g["n_clips"])
        if gkey in g and rank < g[gkey].shape[0] and B >= n:
            got = logits[:n].cpu().numpy()
            ref = g[gkey][rank]
            err = float(np.max(np.abs(got - ref)))
            same = bool(np.array_equal(got[:, -3:].argmax(1), ref[:, -3:].argmax(1)))
            parity = {"checked": True, "clips": n, "max_abs_logit_diff_vs_oracle_golden": err, "tol": GOLDEN_LOGIT_TOL,
                      "argmax_3C_identical": same}
            if not (err <= GOLDEN_LOGIT_TOL and same):
                raise AssertionError("rank %d: bench logits differ from tests/golden/bench_golden.npz: max |diff| = %g "
                                     "(tol %g), argmax identical: %s" % (rank, err, GOLDEN_LOGIT_TOL, same))
    bad = ranks.sum_over_ranks(0.0 if parity["checked"] else 1.0, dev)  # ranks whose configuration has no golden entry

    ms = {n: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in sampled])) for i, n in enumerate(names)}
    kernels = {}
    for n in ("stft", "median", "features"):
        nbytes = BYTES[n]
        if n == "features" and fuse_l0:  # the two layer-0 partials (2 x 68 x 32 f32) leave instead of the patches
            nbytes += 2 * W_PATCH * 32 * 4 - W_PATCH * FEAT * 4
        if n == "features" and want_lay == 2:  # single kernel: the featuregram is written once and never re-read
            nbytes -= FEAT * T_FRAMES * 4
        gbs = nbytes * B / (ms[n] * 1e-3) / 1e9
        kernels[n] = {"ms": round(ms[n], 4), "bound": "hbm", "algorithmic_bytes_per_clip": nbytes,
                      "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    mfma_peak = MFMA_F32_PEAK_TFLOPS if args.model_dtype == "f32" else MFMA_BF16_PEAK_TFLOPS
    flops_model = FLOPS_MODEL - (2.0 * W_PATCH * FEAT * 32 if fuse_l0 else 0.0)  # layer 0 runs in the feature kernel when fused
    if args.classes == 5:  # Dense-on-trunk outputs: 5 + 4 x 16 instead of 3 + 3 x 16
        flops_model += 2.0 * W_PATCH * 32 * (69 - 51)
    tf = flops_model * B / (ms["model"] * 1e-3) / 1e12
    kernels["model"] = {"ms": round(ms["model"], 4), "bound": "mfma", "algorithmic_flops_per_clip": flops_model,
                        "achieved_TFLOPs": round(tf, 2), "frac": round(tf / mfma_peak, 4),
                        "note": "algorithmic FLOPs count the zero-padding taps of the large dilations, which the kernel skips"}
    # measured HBM traffic per launch from the committed rocprofv3 PMC passes (tools/gpu/collect_profiles.sh)
    pmc = {}
    try:
        import glob
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if fs:
            pmc = json.load(open(fs[-1]))
    except Exception:  # no profile committed yet
        pmc = {}

    def traffic(keys):
        vals = [pmc.get(k, {}).get("hbm_bytes_per_launch") for k in keys]
        return None if any(v is None for v in vals) else float(sum(vals))
    feat_keys = (["features_half"] if hp.T % 2 == 0 else ["features_clip"]) if want_lay == 2 else ["hp_feat", "std_patch"]
    for n, keys in (("stft", ["stft"]), ("median", ["median"]), ("features", feat_keys), ("model", ["model"])):
        kernels[n]["pmc_hbm_bytes_per_launch_at_B1024"] = traffic(keys)
    if pmc.get("model", {}).get("SQ_INSTS_MFMA") and B == 1024 and args.model_dtype == "f32":
        issued = pmc["model"]["SQ_INSTS_MFMA"] * 2048.0  # v_mfma_f32_16x16x4_f32: 16*16*4*2 FLOP per wave instruction
        kernels["model"]["issued_TFLOPs_from_SQ_INSTS_MFMA"] = round(issued / (ms["model"] * 1e-3) / 1e12, 2)
        kernels["model"]["issued_frac"] = round(issued / (ms["model"] * 1e-3) / 1e12 / mfma_peak, 4)
    # the widened row in front of the path, measured separately: NOT part of `value`
    from sm_hpss_mtl_amd import silence as _sil
    for _ in range(2):
        _sil.preprocess_signal(audio, 16000, 25, 10)
    pe = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pe[0].record()
    for _ in range(10):
        _sil.preprocess_signal(audio, 16000, 25, 10)
    pe[1].record()
    torch.cuda.synchronize()
    pms = pe[0].elapsed_time(pe[1]) / 10
    pgbs = 8.0 * audio.shape[1] * B / (pms * 1e-3) / 1e9  # read the clip once, write it once
    kernels["preprocess_signal"] = {"ms": round(pms, 4), "bound": "hbm", "achieved_GBs": round(pgbs, 1),
                                    "frac": round(pgbs / HBM_PEAK_GBS, 4), "in_value": False}
    dominant = max(names, key=lambda n: ms[n])
    if dominant == "model":
        roof = {"kernel": "b3mtl_forward_kernel" if args.model_dtype == "f32" else "b3mtl_forward_bf16_kernel", "bound": "mfma",
                "achieved": round(tf, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(tf / mfma_peak, 4),
                "traffic": traffic(["model"]) if (B == 1024 and args.model_dtype == "f32") else None}
    else:
        kn = {"stft": "stft400_kernel", "median": "hpss_median_split_kernel",
              "features": ("features_half_kernel" if hp.T % 2 == 0 else "features_clip_kernel") if want_lay == 2 else "hp_feat_walk_kernel+std_patch_kernel"}[dominant]
        roof = {"kernel": kn, "bound": "hbm", "achieved": kernels[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dominant]["frac"],
                "traffic": traffic({"stft": ["stft"], "median": ["median"], "features": feat_keys}[dominant]) if B == 1024 else None}

    if rank == 0:
        clips_total = world * B * args.steps
        res = {
            "metric": "clips/sec HPSS+MTL-CNN fwd (1s@16kHz)", "value": round(clips_total / elapsed, 1), "unit": "clips/s",
            "n_gpus": world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup, "preroll_steps": max(0, args.preroll),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.model_dtype == "f32" else "f32 front end + bf16 network operands", "data": "synthetic",
            "config": {"workload": "%d x 1s@16kHz clips per GPU: STFT(400/160) -> HPSS %dx%d median + soft mask -> logmel(120) "
                                   "-> standardise -> patch W=68 -> B3_MTL(%d-class) forward" % (B, args.l_harm, args.l_perc, args.classes),
                       "clips_per_gpu": B, "layer0_fused_into_features": bool(fuse_l0), "harm_layout": int(hp.layout),
                       "l_harm": args.l_harm, "l_perc": args.l_perc, "patch": W_PATCH, "sharding": "per-clip, no data-path collective"},
            "roofline": roof, "kernels": kernels,
            "hbm_roofline_pct_median_kernel": round(100 * kernels["median"]["frac"], 2),
            "parity": dict(parity, ranks_without_golden_check=int(round(bad))),
        }
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
    ranks.close()


if __name__ == "__main__":
    main()
