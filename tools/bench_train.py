"""End-to-end TRAINING throughput (BASELINE config 4's shape): per step and per GPU, `--batch` synthetic 1 s clips
(music | speech | mixtures, labels cycling, SMR -5..20 dB) -> HIP front end (STFT -> HPSS 21x11 -> log-mel ->
standardise -> W=68 patches, with the reference's Gaussian noise augmentation) -> B3_MTL training step (training forward,
losses, backward, ONE all-reduce of the flat gradient + BatchNorm batch statistics over RCCL when WORLD_SIZE > 1, SGD with
momentum and clipnorm).
Launch like bench.py:  python tools/bench_train.py [--gpus N]     (starts its own N ranks; the parent never touches the GPU)
                       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py --gpus N
Prints one JSON line on rank 0.  Not the headline metric (bench.py is); documents the training path end to end."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # (a cold box runs its first steps slower: see bench.py)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512, help="clips per GPU per step (a multiple of 3)")
    ap.add_argument("--classes", type=int, default=3, choices=[3, 5])
    ap.add_argument("--patch", type=int, default=68, help="patch width W (68; the reference's drivers also use 99 and 249)")
    ap.add_argument("--shift", type=int, default=0, help="patch shift (default: W, 24 for W = 249 as in Proposed_Work_Results.py:724-725)")
    ap.add_argument("--serial", action="store_true", help="front end and training step of a batch back to back on one stream (default from "
                    "256 clips per step on: the front end of batch k + 1 runs on a second HIP stream beside the training step of batch "
                    "k, as the reference's generator workers prepare the next batch while Keras trains on the current one -- "
                    "measured 0.825 -> 0.785 ms per 510-clip step; at 48 clips the step is launch-bound and the second stream costs 2 %)")
    ap.add_argument("--overlap", action="store_true", help="force the two-stream form at any batch size")
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible weight gradients (fixed-point integer atomics, "
                    "smh_trainer_set_deterministic) instead of float atomics: states what determinism costs per step")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="bf16: the training forward on split bf16 operands "
                    "(smh_trainer_set_dtype; BASELINE config 5's \"mixed bf16 CNN\")")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal (SMH_DIST_BACKEND=gloo on a CPU box): no compute")
    args = ap.parse_args()
    from sm_hpss_mtl_amd.launch import init_ranks, spawn_ranks_if_needed, timed_region
    rc = spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    if rc is not None:  # parent of the ranks: never touched the GPU
        sys.exit(rc)
    ranks = init_ranks(args.gpus)
    rank, world = ranks.rank, ranks.world
    if args.dry_run:
        elapsed, ran = timed_region(ranks, lambda k, timed: time.sleep(0.001), args.steps, args.warmup, lambda: None)
        if rank == 0:
            print(json.dumps({"metric": "launcher dry run (no training step executed)", "value": None, "unit": "clips/s",
                              "n_gpus": world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(1e3 * elapsed / args.steps, 4), "dry_run": True, "backend": ranks.backend}))
        ranks.close()
        return
    import numpy as np
    import torch
    from sm_hpss_mtl_amd.batching import make_labels_3class, make_labels_5class
    from sm_hpss_mtl_amd.device_rng import add_normal_noise
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.synth import synth_clips

    groups = 3 if args.classes == 3 else 5
    bs = args.batch // groups
    B = bs * groups
    W = args.patch
    shift = args.shift or (24 if W == 249 else W)
    fe = Frontend(FrontendConfig(l_harm=21, l_perc=11))
    model = B3MTL(n_feat=240, patch_size=W, n_classes=args.classes, TR_STEPS=100, seed=0)  # same seed: identical replicas
    model.deterministic_gradients = bool(args.deterministic)
    model.train_dtype = args.dtype
    audio = torch.from_numpy(synth_clips(B, seed=2000 + rank)).cuda()  # B distinct clips per rank
    smr = np.array([(-5, 0, 5, 10, 15, 20)[i % 6] for i in range(bs)], np.float64)
    lab = make_labels_3class(bs, smr) if args.classes == 3 else make_labels_5class(bs, smr, smr[::-1].copy())
    nP = fe.num_patches(fe.num_frames(audio.shape[1]), W, shift)  # short clips are tiled along time first (preprocessing.py:139-142)
    assert nP >= 1
    lab = {k: np.repeat(v, nP, axis=0) for k, v in lab.items()}
    y = model.pack_targets(lab)
    assert y.shape[0] == B * nP, (y.shape, B, nP)
    serial = args.serial or (B < 256 and not args.overlap)
    outs = [{}, {}]  # two sets of front-end buffers: batch k + 1 is produced while batch k is trained on
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def produce(slot):  # front end + augmentation of one batch into buffer set `slot`, on the current stream
        out = outs[slot]
        res = fe.run(audio, W=W, shift=shift, out=out)
        out.update(fv=res["fv"], patches=res["patches"])
        x = res["patches"]
        # noise_augmentation, in place on the patches this step produced (Proposed_Work_Results.py:239-242; scale drawn from
        # {5e-3, 1e-3, 5e-4, 1e-4} there): one HIP pass (csrc/smh_rng.hip); the seed is fixed here so that no host draw sits in the step
        return add_normal_noise(x, 1e-3, seed=99, out=x)

    def step(timed=False):  # the serial form: also what the stage breakdown below is measured on
        if timed:
            ev[0].record()
        x = produce(0)
        if timed:
            ev[1].record()
        r = model.train_on_batch(x, y)  # includes the gradient all-reduce and the optimiser step
        if timed:
            ev[2].record()
        return r

    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    ready = [torch.cuda.Event() for _ in range(2)]     # buffer set produced (recorded on `side`)
    consumed = [torch.cuda.Event() for _ in range(2)]  # training step on that buffer set finished (recorded on `main`)
    pending = {}

    def produce_ahead(k):  # enqueue batch k's front end on the side stream
        slot = k & 1
        with torch.cuda.stream(side):
            side.wait_event(consumed[slot])  # (a fresh event is complete)
            pending[k] = produce(slot)
            ready[slot].record(side)

    def pipelined_step(k):
        if k not in pending:
            produce_ahead(k)
        produce_ahead(k + 1)  # queued BEFORE this step's kernels: it runs beside them
        main.wait_event(ready[k & 1])
        r = model.train_on_batch(pending.pop(k), y)
        consumed[k & 1].record(main)
        return r

    last = [None]
    counter = [0]

    def one(k, timed):
        if serial:
            last[0] = step()
        else:
            last[0] = pipelined_step(counter[0])
            counter[0] += 1

    dt, ran = timed_region(ranks, one, args.steps, args.warmup, torch.cuda.synchronize, audio.device)
    assert ran == world
    last = last[0]
    step(timed=True)
    torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps({
            "metric": "clips/sec HPSS + B3_MTL training step (1s@16kHz)", "value": round(world * B * args.steps / dt, 1),
            "unit": "clips/s", "n_gpus": world, "ranks_reporting": ran, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "dtype": "f32" if args.dtype == "f32" else "f32 master weights, saved activations, backward and optimiser; forward on split bf16 operands (3 bf16 MFMA per f32 product)",
            "data": "synthetic", "front_end_overlapped_with_previous_step": not serial,
            "deterministic_gradients": bool(args.deterministic),
            "config": {"workload": "%d clips per GPU per step: front end 21x11 -> W=%d patches (%d per clip) -> B3_MTL(%d-class) "
                                            "train step, SGD(momentum 0.9, clipnorm 1)" % (B, W, nP, args.classes),
                                            "gradient_allreduce_bytes": 4 * model.count_params() if world > 1 else 0},
            "stages_ms_serial": {"front_end_and_augmentation": round(ev[0].elapsed_time(ev[1]), 4),
                          "train_step_incl_allreduce_and_host_sync": round(ev[1].elapsed_time(ev[2]), 4)},
            "last_losses": dict(zip(model.metrics_names, [round(float(v), 5) for v in last]))}))
    ranks.close()


if __name__ == "__main__":
    main()
