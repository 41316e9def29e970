#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic 1 s @ 16 kHz clips.

One "step" = one pass of the whole hot path over one batch that is already resident in HBM (`HotPath.step`,
sm_hpss_mtl_amd/pipeline.py -- the same object tests/test_bench_path_gpu.py puts under the oracle):
  STFT -> HPSS medians (l_harm x l_perc) -> soft masks -> mel -> dB -> standardise -> patches (W=68)
       -> B3_MTL forward (logits)
Default workload = BASELINE.json configs[1] (batch 1024 x 1 s clips, 17x17 medians) carried through the network forward,
i.e. the metric "clips/sec HPSS+MTL-CNN fwd".  `--workload` selects the other single-GPU configurations of BASELINE.json,
each printing the same kind of line with its own `roofline`:
  config2   HPSS only: STFT + medians + soft masks + log-mel featuregram (three launches), batch 1024, 17x17
  config3   B3_MTL (3-class) forward on precomputed standardised patches, batch 256 (smh_model_forward_f32)
  config5   config 5's shape on one GPU: 21x11 medians, 5-class network on SPLIT bf16 operands, batch 1024
(config 4, the training step, is tools/bench_train.py.)

Protocol: W untimed warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides -- what the caller
passes is what runs; `value` comes from that region alone.  A device that was idle runs its first ~20 steps ~9 % slower (clock
ramp), so after the timed region the same step is run `--steady-steps` more times and reported as `steady_state` (extra
information, never `value`).  The batch is B DISTINCT clips (sm_hpss_mtl_amd.synth.bench_clips): every step reads B x 64 000
bytes of different audio.

N > 1: one process per GPU.  `python bench.py --gpus N` starts its own N ranks (a child `torch.distributed.run`; this
parent never touches the GPU); under a launcher (RANK/WORLD_SIZE in the environment) the process is a rank.  Every
rank owns its own batch -- clips are independent units, so there is no collective on the data path (weak
scaling); only the timing barrier / MAX all-reduce use RCCL.  Asking for more GPUs than are visible is an error.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step; per-kernel figures for every
stage are in `kernels`: HIP events on the launch stream around every stage, recorded inside the timed region on every
n-th step (`--event-every`; five markers per step cost ~10 us).  After the timed region the logits of each rank's
clips are compared with tests/golden/bench_golden.npz (CPU-oracle logits of those very clips; data only) -- a bench
whose logits do not match the reference arithmetic fails instead of printing a number -- and the model's device error word is
read (smh_model_status).  `cpu_baseline` (rank 0, N = 1, default workload only, before the GPU is touched) times the CPU path
(numpy rfft + scipy.ndimage.median_filter + the numpy restatement, `python -m oracle.cpu_baseline`) on one core and on all
usable cores.
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
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (only for the split-bf16 network)
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
GOLDEN_LOGIT_TOL_SPLIT_BF16 = 3e-4  # split-bf16 network: measured 5-8e-5 from the f32 kernel

WORKLOADS = {
    # name: (batch, l_harm, l_perc, classes, model dtype, what runs)
    "headline": dict(batch=1024, l_harm=17, l_perc=17, classes=3, model_dtype="f32", path="full"),
    "config2": dict(batch=1024, l_harm=17, l_perc=17, classes=3, model_dtype="f32", path="frontend"),
    "config3": dict(batch=256, l_harm=21, l_perc=11, classes=3, model_dtype="f32", path="model"),
    "config5": dict(batch=1024, l_harm=21, l_perc=11, classes=5, model_dtype="bf16", path="full"),
}


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
    # defaults: 50 + 200 steps = 0.1 s of GPU time (a cold box runs its first ~20 steps 9 % slower: clock ramp)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="steps run AFTER the timed region and reported as steady_state (never as value); 0 = skip")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="headline")
    ap.add_argument("--batch", type=int, default=None, help="clips (config3: patches) per GPU per step; default: the workload's")
    ap.add_argument("--l-harm", type=int, default=None)
    ap.add_argument("--l-perc", type=int, default=None)
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
    ap.add_argument("--classes", type=int, choices=[3, 5], default=None,
                    help="3: the headline B3_MTL (S, M, R, 3C); 5: the musan_5_class variant (S, M, N, R, 5C) of BASELINE config 5")
    ap.add_argument("--model-dtype", choices=["f32", "bf16"], default=None,
                    help="bf16 = SPLIT bf16 operands (hi + lo pairs, three bf16 products per f32 product: f32-grade arithmetic on "
                         "the bf16 matrix pipe, BASELINE config 5); NOT the parity path, never the default")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal: start the ranks, join the process group (SMH_DIST_BACKEND, e.g. gloo on a CPU "
                         "box), run the timing protocol around an empty step and report value = null.  Measures nothing.")
    args = ap.parse_args(argv)
    wl = WORKLOADS[args.workload]
    for k in ("batch", "l_harm", "l_perc", "classes", "model_dtype"):
        if getattr(args, k) is None:
            setattr(args, k, wl[k])
    args.path = wl["path"]
    return args


def dry_run(args, ranks):
    from sm_hpss_mtl_amd.launch import timed_region
    elapsed, ran = timed_region(ranks, lambda k, timed: time.sleep(0.001), args.steps, args.warmup, lambda: None)
    if ranks.rank == 0:
        print(json.dumps({"metric": "launcher dry run (no hot path executed)", "value": None, "unit": "clips/s",
                          "n_gpus": ranks.world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 4), "dry_run": True,
                          "backend": ranks.backend}))
    ranks.close()


def load_pmc():
    """measured HBM traffic / instruction counts per launch from the committed rocprofv3 PMC passes (tools/gpu/collect_profiles.sh)"""
    try:
        import glob
        fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        return json.load(open(fs[-1])) if fs else {}
    except Exception:  # no profile committed yet
        return {}


def main():
    args = parse_args()
    from sm_hpss_mtl_amd.launch import init_ranks, spawn_ranks_if_needed, timed_region
    rc = spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:  # this process was the parent of the ranks: it never touched the GPU
        sys.exit(rc)

    # CPU baseline first (rank 0 of a one-GPU job): a child process, nothing of it overlaps the timed region
    cpu = None
    if (args.gpus == 1 and args.workload == "headline" and not args.no_cpu_baseline and not args.dry_run
            and int(os.environ.get("RANK", "0")) == 0):
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
    from sm_hpss_mtl_amd.synth import bench_clips, synth_clips

    B = args.batch
    # weak scaling: the job is world*B clips; rank r owns the contiguous index range shard_range(...)
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    fe = Frontend(FrontendConfig(l_harm=args.l_harm, l_perc=args.l_perc))
    model = None if args.path == "frontend" else B3MTL(n_feat=FEAT, patch_size=W_PATCH, n_classes=args.classes, seed=0)
    names = list(STAGES)
    pmc = load_pmc()

    if args.path == "model":
        # BASELINE config 3: the network alone on PRECOMPUTED standardised patches (clips of seed 2 through the
        # front end once, outside the timed region), the reference's call structure: model.predict(patches)
        clips = synth_clips(B, seed=2 + 100 * rank)
        patches = fe.run(torch.from_numpy(clips).cuda(), W=W_PATCH, shift=W_PATCH)["patches"]
        assert patches.shape == (B, W_PATCH, FEAT)
        dev = patches.device
        logits = torch.empty((B, model.out_dim), dtype=torch.float32, device=dev)
        names = ["model"]
        hp = None
        fuse_l0, want_lay = False, None
    else:
        audio = torch.from_numpy(bench_clips(B, rank)).cuda()
        dev = audio.device
        hp = HotPath(fe, model, B, audio.shape[1], patch=W_PATCH, fuse_l0=not args.no_fuse_l0,
                     two_kernel_features=args.two_kernel_features, model_dtype=args.model_dtype)
        fuse_l0, want_lay = hp.fuse_l0, hp.want_layout
        if args.path == "frontend":
            names = names[:3]

    # HIP events on the launch stream (torch's current stream IS the stream every kernel is launched on)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    every = max(1, min(args.event_every, args.steps // 5))  # at least five samples when the caller asks for few steps
    sampled = [k for k in range(args.steps) if k % every == every - 1]
    sampled_set = set(sampled)

    def step(k, timed):
        rec = ev[k] if (timed and k in sampled_set) else None
        if hp is not None:
            hp.step(audio, rec)
        else:
            if rec is not None:
                rec[0].record()
            model.forward_device(patches, out=logits, dtype=args.model_dtype)
            if rec is not None:
                rec[1].record()

    elapsed, ran = timed_region(ranks, step, args.steps, args.warmup, torch.cuda.synchronize, dev)
    assert ran == world, "%d ranks ran the timed region, expected %d" % (ran, world)
    if hp is not None:
        logits = hp.logits
    if model is not None:
        model.check_status()  # a kernel that gave up (smh_model_status) fails the bench instead of handing back zeros
        assert torch.isfinite(logits).all(), "non-finite logits"

    # steady state (after the timed region, never part of `value`)
    steady = None
    if args.steady_steps > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steady_steps):
            step(k, False)
        torch.cuda.synchronize()
        st_el = ranks.max_over_ranks(time.perf_counter() - t0, dev)
        steady = {"steps": args.steady_steps, "ms_per_step": round(1e3 * st_el / args.steady_steps, 4),
                  "value": round(world * B * args.steady_steps / st_el, 1),
                  "note": "the same step run again after the timed region (device at its working clock); NOT `value`"}

    # ---- parity of the timed configuration: this rank's clips against the committed oracle logits ----
    parity = {"checked": False}
    gpath = os.path.join(ROOT, "tests", "golden", "bench_golden.npz")
    if os.path.exists(gpath) and model is not None:
        g = np.load(gpath)
        tol = GOLDEN_LOGIT_TOL if args.model_dtype == "f32" else GOLDEN_LOGIT_TOL_SPLIT_BF16
        checks = []  # (golden rows, batch rows)
        if args.path == "model":
            if args.classes == 3 and (args.l_harm, args.l_perc) == (21, 11) and rank == 0 and B >= 8:
                checks.append((g["config3_logits"], slice(0, 8)))
        else:
            gkey = ("logits_%dx%d" if args.classes == 3 else "logits5_%dx%d") % (args.l_harm, args.l_perc)
            n = int(g["n_clips"])
            if gkey in g and rank < g[gkey].shape[0] and B >= n:
                checks.append((g[gkey][rank], slice(0, n)))
            tkey = "logits_tail_%dx%d" % (args.l_harm, args.l_perc)
            r0 = int(g["tail_row"]) if "tail_row" in g else 64
            if args.classes == 3 and tkey in g and rank < g[tkey].shape[0] and B >= r0 + int(g["n_tail"]):
                checks.append((g[tkey][rank], slice(r0, r0 + int(g["n_tail"]))))
        if checks:
            err, same, nchk = 0.0, True, 0
            for ref, rows in checks:
                got = logits[rows].cpu().numpy()
                err = max(err, float(np.max(np.abs(got - ref))))
                same = same and bool(np.array_equal(got[:, -args.classes:].argmax(1), ref[:, -args.classes:].argmax(1)))
                nchk += ref.shape[0]
            parity = {"checked": True, "clips": nchk, "max_abs_logit_diff_vs_oracle_golden": err, "tol": tol,
                      "argmax_identical": same}
            if not (err <= tol and same):
                raise AssertionError("rank %d: bench logits differ from tests/golden/bench_golden.npz: max |diff| = %g "
                                     "(tol %g), argmax identical: %s" % (rank, err, tol, same))
    elif model is None:
        parity = {"checked": False, "note": "front end only: its parity is tests/test_bench_path_gpu.py (medians bit-exact, "
                                            "featuregram 1e-3 dB from the device's S at this batch size)"}
    bad = ranks.sum_over_ranks(0.0 if (parity["checked"] or model is None) else 1.0, dev)  # ranks without a golden entry

    ms = {n: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in sampled])) for i, n in enumerate(names)}
    kernels = {}
    for n in ("stft", "median", "features"):
        if n not in ms:
            continue
        nbytes = BYTES[n]
        if n == "features" and fuse_l0:  # the two layer-0 partials (2 x 68 x 32 f32) leave instead of the patches
            nbytes += 2 * W_PATCH * 32 * 4 - W_PATCH * FEAT * 4
        if n == "features" and model is None:  # front end only: no patches leave
            nbytes -= W_PATCH * FEAT * 4
        if n == "features" and want_lay == 2:  # single kernel: the featuregram is written once and never re-read
            nbytes -= FEAT * T_FRAMES * 4
        gbs = nbytes * B / (ms[n] * 1e-3) / 1e9
        kernels[n] = {"ms": round(ms[n], 4), "bound": "hbm", "algorithmic_bytes_per_clip": nbytes,
                      "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

    def traffic(keys):
        vals = [pmc.get(k, {}).get("hbm_bytes_per_launch") for k in keys]
        return None if (B != 1024 or any(v is None for v in vals)) else float(sum(vals))
    feat_keys = []
    if hp is not None:
        feat_keys = (["features_half"] if hp.T % 2 == 0 else ["features_clip"]) if want_lay == 2 else ["hp_feat", "std_patch"]
    tf = None
    if "model" in ms:
        mfma_peak = MFMA_F32_PEAK_TFLOPS if args.model_dtype == "f32" else MFMA_BF16_PEAK_TFLOPS
        flops_model = FLOPS_MODEL - (2.0 * W_PATCH * FEAT * 32 if fuse_l0 else 0.0)  # layer 0 runs in the feature kernel when fused
        if args.classes == 5:  # Dense-on-trunk outputs: 5 + 4 x 16 instead of 3 + 3 x 16
            flops_model += 2.0 * W_PATCH * 32 * (69 - 51)
        tf = flops_model * B / (ms["model"] * 1e-3) / 1e12
        kernels["model"] = {"ms": round(ms["model"], 4), "bound": "mfma", "algorithmic_flops_per_clip": flops_model,
                            "achieved_TFLOPs": round(tf, 2), "frac": round(tf / mfma_peak, 4),
                            "note": "algorithmic FLOPs (SURVEY 8d) count the zero-padding taps of the large dilations, which the "
                                    "kernel skips; frac_issued counts only the products the matrix cores were given"}
        if args.model_dtype == "bf16":
            # three bf16 products per f32 product: the pipe is given 3 x the algorithmic FLOPs of the trunk
            kernels["model"]["note"] = ("split bf16 operands: every f32 product is three bf16 MFMA products (hi*hi + hi*lo + lo*hi); "
                                        "frac = ALGORITHMIC f32 FLOPs / bf16 peak, frac_issued = 3 x that")
            kernels["model"]["frac_issued"] = round(3.0 * tf / mfma_peak, 4)
        elif pmc.get("model", {}).get("SQ_INSTS_MFMA") and B == 1024 and hp is not None and fuse_l0:
            issued = pmc["model"]["SQ_INSTS_MFMA"] * float(pmc["model"].get("flop_per_mfma", 2048.0))  # 16x16x4 f32: 2048 FLOP
            kernels["model"]["issued_TFLOPs_from_SQ_INSTS_MFMA"] = round(issued / (ms["model"] * 1e-3) / 1e12, 2)
            kernels["model"]["frac_issued"] = round(issued / (ms["model"] * 1e-3) / 1e12 / mfma_peak, 4)
    for n, keys in (("stft", ["stft"]), ("median", ["median"]), ("features", feat_keys), ("model", ["model"])):
        if n in kernels and hp is not None and args.workload == "headline":
            kernels[n]["pmc_hbm_bytes_per_launch_at_B1024"] = traffic(keys)
    if hp is not None and args.workload == "headline":
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
                "frac_issued": kernels["model"].get("frac_issued"),
                "traffic": traffic(["model"]) if (args.workload == "headline" and args.model_dtype == "f32") else None}
    else:
        kn = {"stft": "stft400_kernel", "median": "hpss_median_split_kernel",
              "features": ("features_half_kernel" if hp.T % 2 == 0 else "features_clip_kernel") if want_lay == 2 else "hp_feat_walk_kernel+std_patch_kernel"}[dominant]
        roof = {"kernel": kn, "bound": "hbm", "achieved": kernels[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dominant]["frac"],
                "traffic": traffic({"stft": ["stft"], "median": ["median"], "features": feat_keys}[dominant]) if args.workload == "headline" else None}

    if rank == 0:
        clips_total = world * B * args.steps
        unit_name = "patches" if args.path == "model" else "clips"
        metric = {"full": "clips/sec HPSS+MTL-CNN fwd (1s@16kHz)", "frontend": "clips/sec HPSS front end (STFT + medians + masks + logmel), 1s@16kHz",
                  "model": "patches/sec B3_MTL forward on precomputed patches (W=68, 240 features)"}[args.path]
        what = {"full": "%d x 1s@16kHz clips per GPU: STFT(400/160) -> HPSS %dx%d median + soft mask -> logmel(120) -> standardise -> patch W=68 "
                        "-> B3_MTL(%d-class) forward" % (B, args.l_harm, args.l_perc, args.classes),
                "frontend": "BASELINE config 2: %d x 1s@16kHz clips per GPU: STFT(400/160) -> HPSS %dx%d median + soft mask -> logmel(120) "
                            "featuregram; no patches, no network" % (B, args.l_harm, args.l_perc),
                "model": "BASELINE config 3: B3_MTL(%d-class) forward on %d precomputed standardised patches (68 x 240) per GPU, "
                         "layer 0 included (smh_model_forward_f32)" % (args.classes, B)}[args.path]
        if args.workload == "config5":
            what = "BASELINE config 5 on one GPU: " + what
        dtype = "f32" if args.model_dtype == "f32" else "f32 front end + f32-grade network on split bf16 operands (3 bf16 MFMA products per f32 product)"
        res = {
            "metric": metric, "value": round(clips_total / elapsed, 1), "unit": unit_name + "/s",
            "n_gpus": world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic (%d distinct %s per GPU)" % (B, unit_name),
            "config": {"workload": what, "name": args.workload, unit_name + "_per_gpu": B,
                       "layer0_fused_into_features": bool(fuse_l0), "harm_layout": None if hp is None else int(hp.layout),
                       "l_harm": args.l_harm, "l_perc": args.l_perc, "patch": W_PATCH, "sharding": "per-%s, no data-path collective" % unit_name[:-1]},
            "roofline": roof, "kernels": kernels,
            "parity": dict(parity, ranks_without_golden_check=int(round(bad))),
        }
        if steady is not None:
            res["steady_state"] = steady
        if ranks.dist is not None:  # the backend the barrier / MAX / SUM of the timed region went through
            res["dist_backend"] = ranks.backend
        if "median" in kernels:
            res["hbm_roofline_pct_median_kernel"] = round(100 * kernels["median"]["frac"], 2)
        if cpu is not None:
            res["cpu_baseline"] = cpu
        print(json.dumps(res))
    ranks.close()


if __name__ == "__main__":
    main()
