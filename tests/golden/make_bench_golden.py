"""Generate tests/golden/bench_golden.npz (run in the BUILD container: python tests/golden/make_bench_golden.py).

Oracle logits of the very clips `bench.py` processes: rank r's clips are `synth_clips(64, seed=1000 + r)` and the
network is `B3MTL(seed=0)` = `sm_hpss_mtl_amd.model.initial_weights(seed=0)` (host-only numpy).  For every rank
0..7 the first N_CLIPS clips go through the whole CPU oracle chain
    stft_mag -> hpss (l_harm x l_perc medians + soft masks) -> mel -> power_to_db -> StandardScaler per half
             -> patches (W = 68, shift 68) -> B3_MTL forward
and the concatenated outputs [S | M | R | 3C] are stored.  bench.py compares its own logits of those clips against
this file (data only) instead of merely checking that they are finite; tests/test_bench_path_gpu.py does the same
at B = 1024 and additionally runs the oracle live.
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
W = 68


def oracle_logits(y, w, lh, lp):
    fv = ofe.featuregram(y, "LogMelHarmPercSpec", l_harm=lh, l_perc=lp)
    x = ofe.tcn_input(ofe.feature_patches(fv, W, W))
    return np.concatenate(b3_mtl.forward(x, w), axis=1)[0]


def main():
    _, w = initial_weights(240, W, 3, seed=0)
    g = {"n_clips": np.array(N_CLIPS), "model_seed": np.array(0), "clip_seed_base": np.array(1000)}
    for lh, lp in ((17, 17), (21, 11)):
        out = np.empty((8, N_CLIPS, 7), np.float32)
        for r in range(8):
            clips = synth_clips(64, seed=1000 + r)[:N_CLIPS]
            for i in range(N_CLIPS):
                out[r, i] = oracle_logits(clips[i], w, lh, lp)
        g["logits_%dx%d" % (lh, lp)] = out
    np.savez_compressed(os.path.join(OUT, "bench_golden.npz"), **g)
    print("bench_golden.npz", os.path.getsize(os.path.join(OUT, "bench_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
