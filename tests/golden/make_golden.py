"""Generate the committed golden fixtures (run in the BUILD container only: python tests/golden/make_golden.py).

Boundary oracles used here (the exact routines the reference delegates to):
  scipy.ndimage.median_filter(mode='reflect')         <- librosa.decompose.hpss
  sklearn.preprocessing.StandardScaler                 <- lib/preprocessing.py:211-214
  oracle/_ref/tools*.so = the reference's own lib/cython_impl/tools.pyx compiled unmodified
                                                       <- tools.extract_patches
Everything librosa/keras-side comes from the numpy restatement in oracle/ ("parity unpinned").
Fixtures are data only (inputs are re-creatable from the seeded generator; outputs are stored).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

from scipy.ndimage import median_filter  # noqa: E402
from sklearn.preprocessing import StandardScaler  # noqa: E402

import tools as ref_tools  # noqa: E402  (compiled reference module)
from oracle import b3_mtl, frontend as ofe  # noqa: E402
from sm_hpss_mtl_amd.synth import synth_clips  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def checks(a):
    a = np.asarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()])


def main():
    clips = synth_clips(4, seed=0)
    y = clips[0]
    # ---- front end, clip 0, the reference's configuration (21, 11) ----
    S = ofe.stft_mag(y)
    harm = median_filter(S, size=(1, 21), mode="reflect")
    perc = median_filter(S, size=(11, 1), mode="reflect")
    fv, parts = ofe.featuregram(y, "LogMelHarmPercSpec", l_harm=21, l_perc=11, return_parts=True)
    assert np.array_equal(parts["harm"], harm) and np.array_equal(parts["perc"], perc), "oracle median != scipy"
    half = fv.shape[0] // 2
    stdH = StandardScaler(copy=True).fit_transform(fv[:half].T.copy()).T
    stdP = StandardScaler(copy=True).fit_transform(fv[half:].T.copy()).T
    g = dict(audio0_check=checks(y), S=S, harm_21=harm, perc_11=perc, H=parts["H"], P=parts["P"], fv=fv,
             std_H=stdH.astype(np.float32), std_P=stdP.astype(np.float32),
             mel_basis=ofe.mel_basis(22050, 400, 120))
    # BASELINE config 2 window (17,17): rows/cols subsample + checksums
    h17 = median_filter(S, size=(1, 17), mode="reflect")
    p17 = median_filter(S, size=(17, 1), mode="reflect")
    g.update(harm_17_rows=h17[::25], perc_17_rows=p17[::25], harm_17_check=checks(h17), perc_17_check=checks(p17))
    # other clips: checksums of every stage
    for i in range(1, 4):
        fvi, pi = ofe.featuregram(clips[i], "LogMelHarmPercSpec", return_parts=True)
        assert np.array_equal(pi["harm"], median_filter(pi["S"], size=(1, 21), mode="reflect"))
        assert np.array_equal(pi["perc"], median_filter(pi["S"], size=(11, 1), mode="reflect"))
        g["clip%d_check" % i] = np.stack([checks(pi[k]) for k in ("S", "harm", "perc", "H", "P")] + [checks(fvi)])
    # ---- patches from the compiled reference module, incl. tile-if-short geometry ----
    for W, shift in ((68, 68), (68, 34), (99, 34), (249, 24)):
        FVt = ofe.tile_if_short(fv[:half], W)
        pr = ref_tools.extract_patches(FVt, FVt.shape, W, shift)
        g["patches_W%d_s%d_shape" % (W, shift)] = np.array(pr.shape)
        g["patches_W%d_s%d_starts" % (W, shift)] = np.array(ofe.patch_starts(FVt.shape[1], W, shift))
        g["patches_W%d_s%d_check" % (W, shift)] = checks(pr)
        if W == 68 and shift == 34:
            g["patches_W68_s34_first8rows"] = pr[:, :8, :].astype(np.float32)
    # integer contract table: (T, W, shift) -> nP from the compiled reference
    tbl = []
    for T in (1, 33, 34, 35, 67, 68, 69, 98, 99, 100, 136, 196, 249, 250, 294, 1000):
        for W in (68, 99, 249, 5, 4):
            for shift in (1, 24, 34, 68):
                FVx = np.zeros((2, T), np.float32)
                tbl.append((T, W, shift, ref_tools.extract_patches(FVx, FVx.shape, W, shift).shape[0]))
    g["npatch_table"] = np.array(tbl, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "frontend_golden.npz"), **g)

    # ---- B3_MTL logits from the numpy restatement with seeded weights ----
    m = {}
    for ncls, W in ((3, 68), (5, 68), (3, 99)):
        w = b3_mtl.init_weights(seed=7, n_feat=240, patch_size=W, n_classes=ncls, randomize_bn=True)
        rng = np.random.default_rng(11)
        x = rng.standard_normal((6, W, 240)).astype(np.float32)
        outs, trunk = b3_mtl.forward(x, w, n_classes=ncls, return_trunk=True)
        m["out_c%d_W%d" % (ncls, W)] = np.concatenate(outs, axis=1)
        m["trunk_check_c%d_W%d" % (ncls, W)] = checks(trunk)
        m["wcheck_c%d_W%d" % (ncls, W)] = checks(np.concatenate([v.ravel() for v in w.values()]))
    np.savez_compressed(os.path.join(OUT, "b3mtl_golden.npz"), **m)
    for f in ("frontend_golden.npz", "b3mtl_golden.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
