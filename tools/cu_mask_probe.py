"""What could a mixed-stage grid gain?  A bound from the kernels as they are.

The four stages of the hot path (STFT, HPSS medians, features + layer 0, network) are full-grid kernels: issued on several streams they
are not co-scheduled (round 3: two streams buy nothing).  hipExtStreamCreateWithCUMask can force it: each stage gets its own stream
restricted to some of the eight XCDs, and the four stages then run AT THE SAME TIME on different batches -- the software-pipelined steady state a persistent
mixed-stage grid would reach, minus the overlap INSIDE a CU.  If memory phases of one stage really can hide under the arithmetic of
another at the level of the chip (HBM, fabric), the concurrent round is shorter than the sequential one; if every stage simply slows
down in proportion to the CUs it lost, it is not.

    python tools/cu_mask_probe.py [--rounds 200]          (one MI355X; prints a JSON summary)

Inputs of every stage are the outputs of one ordinary pass over the bench's clips, so each stage does its real work every round.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024)
    args = ap.parse_args()
    import torch

    from sm_hpss_mtl_amd import _lib
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.model import B3MTL
    from sm_hpss_mtl_amd.pipeline import HotPath, _p
    from sm_hpss_mtl_amd.synth import bench_clips

    hip = C.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32

    def masked_stream(lo, hi, bits=None):  # CU mask bits [lo, hi), or an explicit list
        mask = (C.c_uint32 * words)()
        for b in (range(lo, hi) if bits is None else bits):
            mask[b // 32] |= 1 << (b % 32)
        s = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
        if rc != 0:
            raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
        return torch.cuda.ExternalStream(s.value)

    B = args.batch
    fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
    model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
    audio = torch.from_numpy(bench_clips(B, 0)).cuda()
    hp = HotPath(fe, model, B, audio.shape[1], patch=68)
    for _ in range(20):
        hp.step(audio)
    torch.cuda.synchronize()
    lib, h = hp.lib, hp._h
    lay = hp.layout

    def stage(i):  # enqueue stage i on torch's current stream, on the buffers of the ordinary pass
        st = _lib.current_stream()
        if i == 0:
            _lib.check(lib.smh_stft_mag_f32(h, _p(audio), B, hp.n_samples, _p(hp.S), st), "stft")
        elif i == 1:
            _lib.check(lib.smh_hpss_median_ex_f32(h, _p(hp.S), B, fe.K, hp.T, 17, 17, _p(hp.harm), _p(hp.perc), hp.want_layout, st), "median")
        elif i == 2:
            _lib.check(lib.smh_features_l0_f32(h, _p(hp.S), _p(hp.harm), _p(hp.perc), lay, B, hp.T, hp.W, hp.shift, _p(hp.fv), None,
                                               C.c_void_p(lib.smh_model_w0_ptr(model._h)), _p(hp.x0p), _p(hp.maxkeys), st), "features")
        else:
            model.forward_from_x0(hp.x0p, out=hp.logits)

    R = args.rounds

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / R * 1e6  # us per round

    out = {"n_cu": n_cu, "rounds": R, "batch": B}
    # 1. the step as it is: the four stages in sequence on one unrestricted stream
    out["sequential_full_chip_us"] = timed(lambda: [stage(i) for _ in range(R) for i in range(4)])
    alone_full = [timed(lambda i=i: [stage(i) for _ in range(R)]) for i in range(4)]
    out["stage_alone_full_chip_us"] = alone_full
    # 1b. how the mask's bits map to CUs: the same number of bits, contiguous against strided -- a stage whose workgroups are dealt
    # round-robin over the XCDs runs at the pace of the XCD with the fewest enabled CUs
    cal = {}
    for n in (32, 64, 128):
        for kind, bits in (("contiguous", list(range(n))), ("strided", list(range(0, n_cu, n_cu // n)))):
            st_ = masked_stream(0, 0, bits)
            row = []
            for i in (0, 3):
                def run(i=i):
                    with torch.cuda.stream(st_):
                        for _ in range(R):
                            stage(i)
                row.append(timed(run))
            cal["%d_bits_%s" % (n, kind)] = {"stft_us": row[0], "network_us": row[1]}
    out["mask_calibration"] = cal
    # 2. every stage on its own XCDs (the mask's bits are XCD-major in groups of eight CUs -- see the calibration: 32 contiguous bits are
    # one XCD, 32 bits strided by eight enable everything).  A share that cuts an XCD runs at the pace of the cut one (the dispatcher
    # deals a grid's workgroups round-robin over the XCDs), so shares are whole XCDs; 43 : 63 : 96 : 121 us wants 1.1 : 1.6 : 2.4 : 3.0.
    total = sum(alone_full)
    per = n_cu // 8
    for name, xcds in (("xcds_1_2_2_3", (1, 2, 2, 3)), ("xcds_1_1_3_3", (1, 1, 3, 3)), ("xcds_2_2_2_2", (2, 2, 2, 2))):
        cuts = [0]
        for x in xcds:
            cuts.append(cuts[-1] + x * per)
        streams = [masked_stream(cuts[i], cuts[i + 1]) for i in range(4)]
        alone = []
        for i in range(4):
            def run(i=i):
                with torch.cuda.stream(streams[i]):
                    for _ in range(R):
                        stage(i)
            alone.append(timed(run))

        def together():
            for _ in range(R):
                for i in range(4):
                    with torch.cuda.stream(streams[i]):
                        stage(i)
        t_all = timed(together)
        out[name] = {"xcds_per_stage": list(xcds), "stage_alone_on_its_xcds_us": alone,
                     "stage_alone_if_it_scaled_with_its_cus_us": [alone_full[i] * 8 / xcds[i] for i in range(4)],
                     "four_stages_concurrently_us_per_round": t_all, "sum_of_the_stages_alone_on_the_full_chip_us": total}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
