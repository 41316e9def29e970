"""Generate tests/golden/silence_golden.npz (run in the BUILD container: python tests/golden/make_silence_golden.py).

Outputs come from the reference's own lib/cython_impl/tools.pyx compiled unmodified (oracle/_ref/tools*.so;
`tools.medfilt` rebound to a float64-returning scipy.signal.medfilt because modern scipy keeps the integer dtype the
2021 one promoted).  The energies fed to it come from the numpy restatement of librosa.feature.rms (librosa is not
installed here); they are stored so that the GPU test can hand the very same float32 values to both sides.
Inputs are re-creatable from sm_hpss_mtl_amd.synth.gappy_clip; only checksums of them are stored.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))

import scipy.signal as ss  # noqa: E402

import tools as ref_tools  # noqa: E402  (compiled reference module)
from oracle import silence as osil  # noqa: E402
from sm_hpss_mtl_amd.synth import SILENCE_CASES, gappy_clip  # noqa: E402

ref_tools.medfilt = lambda v, k: ss.medfilt(np.asarray(v, float), k)
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    g = {}
    for c in range(len(SILENCE_CASES)):
        x = osil.normalize_signal(gappy_clip(c))
        assert x.dtype == np.float32
        e = osil.rms(x, 400, 160)
        out, sm, fm, tot = ref_tools.removeSilence(x, len(x), e, len(e), 16000, 25, 10)
        o2, s2, f2, t2 = osil.remove_silence(x, e, 16000, 25, 10)
        assert np.array_equal(out, o2) and np.array_equal(sm, s2) and np.array_equal(fm, f2) and tot == t2
        g["c%d_x_sha" % c] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), np.uint8)
        g["c%d_energy" % c] = e
        g["c%d_frame_marker" % c] = fm.astype(np.int8)
        g["c%d_sample_marker" % c] = np.packbits(sm.astype(np.uint8))
        g["c%d_out_sha" % c] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest(), np.uint8)
        g["c%d_meta" % c] = np.array([len(x), int(sm.sum()), int(out is x), tot], np.int64)
    np.savez_compressed(os.path.join(OUT, "silence_golden.npz"), **g)
    print("silence_golden.npz", os.path.getsize(os.path.join(OUT, "silence_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
