"""Generate tests/golden/bench_golden.npz (run in the BUILD container: python tests/golden/make_bench_golden.py).

Oracle logits of the very clips `bench.py` processes: rank r's batch is `synth_clips(64, seed=1000 + r)` followed by
`synth_clips(B - 64, seed=5000 + r)` (all distinct; sm_hpss_mtl_amd.synth.bench_clips) and the network is `B3MTL(seed=0)` =
`sm_hpss_mtl_amd.model.initial_weights(seed=0)` (host-only numpy).  For every rank 0..7 the first N_CLIPS clips and the first
N_TAIL clips of the second part (batch rows 64, 65) go through the whole CPU oracle chain
    stft_mag -> hpss (l_harm x l_perc medians + soft masks) -> mel -> power_to_db -> StandardScaler per half
             -> patches (W = 68, shift 68) -> B3_MTL forward
and the concatenated outputs [S | M | R | 3C] are stored.  bench.py compares its own logits of those clips against
this file (data only) instead of merely checking that they are finite; tests/test_bench_path_gpu.py does the same
at B = 1024 and additionally runs the oracle live.
Also: `logits5_21x11` -- the 5-class network on the same clips (bench.py --workload config5), and `config3_logits` -- the
first 8 of BASELINE config 3's 256 clips (`synth_clips(256, seed=2)`, 21 x 11, 3-class; bench.py --workload config3).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import b3_mtl, frontend as ofe  # noqa: E402
from sm_hpss_mtl_amd.model import initial_weights  # noqa: E402  (host-only numpy)
from sm_hpss_mtl_amd.synth import synth_clips  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_CLIPS = 4
N_TAIL = 2
W = 68


def oracle_logits(y, w, lh, lp, ncls=3):
    fv = ofe.featuregram(y, "LogMelHarmPercSpec", l_harm=lh, l_perc=lp)
    x = ofe.tcn_input(ofe.feature_patches(fv, W, W))
    return np.concatenate(b3_mtl.forward(x, w, n_classes=ncls), axis=1)[0]


def main():
    _, w = initial_weights(240, W, 3, seed=0)
    _, w5 = initial_weights(240, W, 5, seed=0)
    g = {"n_clips": np.array(N_CLIPS), "n_tail": np.array(N_TAIL), "tail_row": np.array(64), "model_seed": np.array(0),
         "clip_seed_base": np.array(1000), "tail_seed_base": np.array(5000)}
    for lh, lp in ((17, 17), (21, 11)):
        out = np.empty((8, N_CLIPS, 7), np.float32)
        tail = np.empty((8, N_TAIL, 7), np.float32)
        for r in range(8):
            clips = synth_clips(64, seed=1000 + r)[:N_CLIPS]
            for i in range(N_CLIPS):
                out[r, i] = oracle_logits(clips[i], w, lh, lp)
            tclips = synth_clips(N_TAIL, seed=5000 + r)
            for i in range(N_TAIL):
                tail[r, i] = oracle_logits(tclips[i], w, lh, lp)
        g["logits_%dx%d" % (lh, lp)] = out
        g["logits_tail_%dx%d" % (lh, lp)] = tail
    out5 = np.empty((8, N_CLIPS, 11), np.float32)
    for r in range(8):
        clips = synth_clips(64, seed=1000 + r)[:N_CLIPS]
        for i in range(N_CLIPS):
            out5[r, i] = oracle_logits(clips[i], w5, 21, 11, 5)
    g["logits5_21x11"] = out5
    c3 = synth_clips(8, seed=2)
    g["config3_logits"] = np.stack([oracle_logits(c3[i], w, 21, 11) for i in range(8)])
    np.savez_compressed(os.path.join(OUT, "bench_golden.npz"), **g)
    print("bench_golden.npz", os.path.getsize(os.path.join(OUT, "bench_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
