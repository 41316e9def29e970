#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic 1 s @ 16 kHz clips.

One "step" = one pass of the whole hot path over one batch that is already resident in HBM:
  STFT -> HPSS medians (l_harm x l_perc) -> soft masks -> mel -> dB -> standardise -> patches (W=68)
       -> B3_MTL forward (logits)
Workload = BASELINE.json configs[1] (batch 1024 x 1 s clips, 17x17 medians) carried through the
network forward, i.e. the metric "clips/sec HPSS+MTL-CNN fwd".  N>1: one process per GPU (launched by
torch.distributed.run), every rank owns its own 1024 clips -- clips are independent units, so there is
no collective on the data path (weak scaling); only the timing barrier/all-reduce uses RCCL.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step; per-kernel
figures for every stage are in `kernels`: HIP events on the launch stream around every stage, recorded inside
the timed region on every 4th step (`--event-every`; five markers per step cost ~10 us).  `cpu_baseline` times the numpy oracle ("port") on a bounded
sample of the same workload on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

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


def cpu_baseline(clips, l_harm, l_perc, seed=0, budget_s=12.0):
    """Oracle ("port": numpy restatement calling the same numpy/scipy-level routines librosa delegates to)
    timed on the host cores of this box, single process, on a bounded sample of the same clips."""
    from oracle import b3_mtl, frontend as ofe  # the checker, used here only as the CPU baseline
    weights = b3_mtl.init_weights(seed=seed)

    def one(y):
        fv = ofe.featuregram(y, "LogMelHarmPercSpec", l_harm=l_harm, l_perc=l_perc)
        x = ofe.tcn_input(ofe.feature_patches(fv, W_PATCH, W_PATCH))
        return b3_mtl.forward(x, weights)

    t0 = time.perf_counter()
    one(clips[0])
    per = time.perf_counter() - t0
    n = int(max(2, min(len(clips), budget_s / max(per, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n):
        one(clips[i])
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "clips/s", "cores": 1, "kind": "port",
            "sample": "%d of the same synthetic clips, numpy oracle front end (%dx%d) + numpy B3_MTL forward, 1 process" % (n, l_harm, l_perc)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="clips per GPU per step")
    ap.add_argument("--l-harm", type=int, default=17)
    ap.add_argument("--l-perc", type=int, default=17)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fuse-l0", action="store_true",
                    help="write standardised patches and let the network read them (the reference's call structure) instead "
                         "of computing the network's first 1x1 convolution inside the feature kernel")
    ap.add_argument("--two-kernel-features", action="store_true",
                    help="time-major harm + hp_feat_walk / std_patch kernels instead of the single feature kernel")
    ap.add_argument("--event-every", type=int, default=4,
                    help="record the per-kernel HIP events on every n-th timed step (the markers of all five stages cost "
                         "about 10 us per step, 2 %% of it; the other timed steps run without them)")
    ap.add_argument("--model-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = mixed-precision network (BASELINE config 5); NOT the parity path, never the default")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.sharding import shard_range
    from sm_hpss_mtl_amd.synth import synth_clips

    B = args.batch
    # weak scaling: the job is world*B clips; rank r owns the contiguous index range shard_range(...)
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    base = synth_clips(64, seed=1000 + rank)  # 64 distinct clips per rank, tiled to the batch
    audio = torch.from_numpy(np.tile(base, ((B + 63) // 64, 1))[:B]).cuda()

    fe = Frontend(FrontendConfig(l_harm=args.l_harm, l_perc=args.l_perc))
    model = B3MTL(n_feat=FEAT, patch_size=W_PATCH, n_classes=3, seed=0)
    T = fe.num_frames(audio.shape[1])
    dev = audio.device
    S = torch.empty((B, fe.K, T), device=dev)
    perc = torch.empty_like(S)
    harm = torch.empty((B, fe.lib.smh_harm_buffer_floats(fe.K, T)), device=dev)  # room for every harm layout
    feat_out = {"fv": torch.empty((B, FEAT, T), device=dev), "patches": torch.empty((B, W_PATCH, FEAT), device=dev),
                "maxkeys": torch.empty(2 * B, dtype=torch.int32, device=dev)}
    logits = torch.empty((B, model.out_dim), device=dev)
    trunk = torch.empty((B, W_PATCH, 32), device=dev)
    import ctypes as C
    from sm_hpss_mtl_amd import _lib
    lib, h = fe.lib, fe._h
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    fuse_l0 = (not args.no_fuse_l0) and args.model_dtype == "f32"
    model._sync_weights()
    w0_ptr = C.c_void_p(lib.smh_model_w0_ptr(model._h))
    x0p = torch.empty((B, 2, W_PATCH, 32), device=dev)
    # harmonic median layout: 16-frame blocks when the single-kernel feature path takes the clip, else time-major
    want_lay = 2 if (lib.smh_features_blocked_ok(h, T, 1 if fuse_l0 else 0) and not args.two_kernel_features) else 1
    names = ["stft", "median", "features", "model"]
    ev = None

    def step(record=None):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if record is not None:
            record[0].record()
        _lib.check(lib.smh_stft_mag_f32(h, p(audio), B, audio.shape[1], p(S), st))
        if record is not None:
            record[1].record()
        lay = _lib.check(lib.smh_hpss_median_ex_f32(h, p(S), B, fe.K, T, args.l_harm, args.l_perc, p(harm), p(perc), want_lay, st))
        if record is not None:
            record[2].record()
        if fuse_l0:
            _lib.check(lib.smh_features_l0_f32(h, p(S), p(harm), p(perc), lay, B, T, W_PATCH, W_PATCH, p(feat_out["fv"]), None,
                                               w0_ptr, p(x0p), p(feat_out["maxkeys"]), st))
        else:
            _lib.check(lib.smh_features_ex_f32(h, p(S), p(harm), p(perc), lay, B, T, W_PATCH, W_PATCH, p(feat_out["fv"]),
                                               p(feat_out["patches"]), p(feat_out["maxkeys"]), st))
        if record is not None:
            record[3].record()
        if fuse_l0:
            model.forward_from_x0(x0p, out=logits, trunk=trunk)
        else:
            model.forward_device(feat_out["patches"], out=logits, trunk=trunk if args.model_dtype == "f32" else None,
                                 dtype=args.model_dtype)
        if record is not None:
            record[4].record()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events on the launch stream (torch's current stream IS the stream every kernel is launched on)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampled = [k for k in range(args.steps) if k % max(1, args.event_every) == 0]
    for k in range(args.steps):
        step(ev[k] if k in sampled else None)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(logits).all(), "non-finite logits"

    ms = {n: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in sampled])) for i, n in enumerate(names)}
    kernels = {}
    for n in ("stft", "median", "features"):
        nbytes = BYTES[n]
        if n == "features" and fuse_l0:  # the two layer-0 partials (2 x 68 x 32 f32) leave instead of the patches
            nbytes += 2 * W_PATCH * 32 * 4 - W_PATCH * FEAT * 4
        if n == "features" and want_lay == 2:  # single kernel: the featuregram is written once and never re-read
            nbytes -= FEAT * T_FRAMES * 4
        gbs = nbytes * B / (ms[n] * 1e-3) / 1e9
        kernels[n] = {"ms": round(ms[n], 4), "bound": "hbm", "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
    mfma_peak = MFMA_F32_PEAK_TFLOPS if args.model_dtype == "f32" else MFMA_BF16_PEAK_TFLOPS
    flops_model = FLOPS_MODEL - (2.0 * W_PATCH * FEAT * 32 if fuse_l0 else 0.0)  # layer 0 runs in the feature kernel when fused
    tf = flops_model * B / (ms["model"] * 1e-3) / 1e12
    kernels["model"] = {"ms": round(ms["model"], 4), "bound": "mfma", "achieved_TFLOPs": round(tf, 2),
                        "frac": round(tf / mfma_peak, 4)}
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
    for n, keys in (("stft", ["stft"]), ("median", ["median"]), ("features", ["features_clip"] if want_lay == 2 else ["hp_feat", "std_patch"]), ("model", ["model"])):
        kernels[n]["pmc_hbm_bytes_per_launch_at_B1024"] = traffic(keys)
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
        kn = {"stft": "stft400_kernel", "median": "hpss_median_split_kernel", "features": "features_clip_kernel" if want_lay == 2 else "hp_feat_walk_kernel+std_patch_kernel"}[dominant]
        roof = {"kernel": kn, "bound": "hbm", "achieved": kernels[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kernels[dominant]["frac"],
                "traffic": traffic({"stft": ["stft"], "median": ["median"], "features": ["features_clip"] if want_lay == 2 else ["hp_feat", "std_patch"]}[dominant]) if B == 1024 else None}

    if rank == 0:
        clips_total = world * B * args.steps
        res = {
            "metric": "clips/sec HPSS+MTL-CNN fwd (1s@16kHz)", "value": round(clips_total / elapsed, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.model_dtype == "f32" else "f32 front end + bf16 network operands", "data": "synthetic",
            "config": {"workload": "%d x 1s@16kHz clips per GPU: STFT(400/160) -> HPSS %dx%d median + soft mask -> logmel(120) "
                                   "-> standardise -> patch W=68 -> B3_MTL(3-class) forward" % (B, args.l_harm, args.l_perc),
                       "clips_per_gpu": B, "layer0_fused_into_features": bool(fuse_l0), "l_harm": args.l_harm, "l_perc": args.l_perc, "patch": W_PATCH, "sharding": "per-clip, no data-path collective"},
            "roofline": roof, "kernels": kernels,
            "hbm_roofline_pct_median_kernel": round(100 * kernels["median"]["frac"], 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(base, args.l_harm, args.l_perc)
            res["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
