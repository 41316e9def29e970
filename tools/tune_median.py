"""GPU tuning harness for the HPSS median kernel: sweeps the segment split (SMH_MEDIAN_SEG) for both
benchmark window pairs in ONE process, interleaved rounds, and prints median/min kernel time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig  # noqa: E402
from sm_hpss_mtl_amd.synth import synth_clips  # noqa: E402

B = 1024
fe = Frontend(FrontendConfig())
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1), (B // 64, 1))).cuda()
S = fe.stft_mag(audio)
harm, perc = torch.empty_like(S), torch.empty_like(S)
BYTES = 3 * 201 * 98 * 4 * B


def time_cfg(lh, lp, seg, rounds=15):
    if seg:
        os.environ["SMH_MEDIAN_SEG"] = seg
    else:
        os.environ.pop("SMH_MEDIAN_SEG", None)
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fe.hpss_median(S, lh, lp)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts = np.array(ts[3:])
    return float(np.median(ts)), float(ts.min())


for lh, lp in ((17, 17), (21, 11)):
    for seg in (None, "1,1", "1,2", "1,3", "2,2", "2,3", "2,4", "3,3", "3,4", "3,5", "4,4", "4,6"):
        try:
            med, mn = time_cfg(lh, lp, seg)
            print("(%d,%d) seg=%-5s  median %.4f ms  min %.4f ms  -> %.1f%% of 8 TB/s" % (lh, lp, seg, med, mn, 100 * BYTES / (med * 1e-3) / 8e12), flush=True)
        except Exception as e:  # plan rejected (e.g. > 16 waves)
            print("(%d,%d) seg=%s rejected: %s" % (lh, lp, seg, str(e)[:80]), flush=True)
