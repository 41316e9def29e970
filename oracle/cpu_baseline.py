"""CPU baseline of the hot path, timed on the host cores (TEST / MEASUREMENT INFRASTRUCTURE, see oracle/__init__).

    python -m oracle.cpu_baseline --l-harm 17 --l-perc 17 --budget 10 [--workers N]

What runs per clip is what the reference's CPU path runs (lib/preprocessing.py:414-424, 137-234; Proposed_Work_Results.py
:483-484, 520), with the third-party routines it delegates to called directly where they exist on this machine:

    numpy.fft.rfft on Hann-windowed frames            <- librosa.core.stft (center=False)
    scipy.ndimage.median_filter(mode='reflect') x 2   <- librosa.decompose.hpss      (~93 % of the front end)
    softmask / mel / power_to_db / StandardScaler / extract_patches / B3_MTL forward: the numpy restatement of oracle/

librosa, TensorFlow and keras-tcn are not installable here, hence kind = "port" (a librosa-equivalent restatement),
not "reference".  If scipy is missing the numpy gather+sort medians of oracle/frontend.py are timed instead and the
`routine` field says so.

Prints ONE JSON object: the single-core figure (BLAS pinned to one thread) and an all-cores figure from a process
pool over clips (clips are independent), each on a bounded sample of `synth_clips(64, seed)` cycled.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W_PATCH = 68
_STATE = {}


def _have_scipy():
    try:
        import scipy.ndimage  # noqa: F401
        return True
    except Exception:
        return False


def _init(l_harm, l_perc, seed, model_seed):
    """Per-process state (also the pool initializer): clips, weights, one BLAS thread."""
    try:
        from threadpoolctl import threadpool_limits
        _STATE["limit"] = threadpool_limits(1)
    except Exception:
        pass
    from oracle import b3_mtl
    from sm_hpss_mtl_amd.synth import synth_clips  # numpy-only generator shared with the GPU path
    _STATE.update(clips=synth_clips(64, seed=seed), w=b3_mtl.init_weights(seed=model_seed), lh=l_harm, lp=l_perc,
                  scipy=_have_scipy())


def one_clip(i):
    from oracle import b3_mtl, frontend as ofe
    y = _STATE["clips"][i % 64]
    lh, lp = _STATE["lh"], _STATE["lp"]
    S = ofe.stft_mag(y)
    if _STATE["scipy"]:
        from scipy.ndimage import median_filter
        harm = median_filter(S, size=(1, lh), mode="reflect")
        perc = median_filter(S, size=(lp, 1), mode="reflect")
    else:
        harm, perc = ofe.median_time(S, lh), ofe.median_freq(S, lp)
    H, P = S * ofe.softmask(harm, perc), S * ofe.softmask(perc, harm)
    fv = np.append(ofe.power_to_db(ofe.mel_project(H, 120) ** 2), ofe.power_to_db(ofe.mel_project(P, 120) ** 2), axis=0)
    x = ofe.tcn_input(ofe.feature_patches(fv.astype(np.float32), W_PATCH, W_PATCH))
    return float(b3_mtl.forward(x, _STATE["w"])[-1][0, 0])


def _chunk(args):
    lo, hi = args
    t0 = time.perf_counter()
    for i in range(lo, hi):
        one_clip(i)
    return time.perf_counter() - t0


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def measure(l_harm, l_perc, budget_s=10.0, workers=None, seed=1000, model_seed=0, max_workers=64):
    _init(l_harm, l_perc, seed, model_seed)
    routine = ("numpy.fft.rfft + scipy.ndimage.median_filter(mode='reflect') + numpy softmask/mel/dB/StandardScaler/"
               "patches + numpy B3_MTL forward") if _STATE["scipy"] else \
              "numpy.fft.rfft + numpy gather/sort medians (scipy missing) + numpy mel/dB/patches + numpy B3_MTL forward"
    one_clip(0)  # warm: mel basis, BLAS
    t0 = time.perf_counter()
    one_clip(1)
    per = max(time.perf_counter() - t0, 1e-4)
    n1 = int(max(4, min(4096, budget_s / per)))
    t0 = time.perf_counter()
    for i in range(n1):
        one_clip(i)
    dt1 = time.perf_counter() - t0
    single = n1 / dt1
    res = {"value": round(single, 2), "unit": "clips/s", "cores": 1, "kind": "port", "routine": routine,
           "sample": "%d of the bench's synthetic 1 s clips (seed %d, cycled), %dx%d medians, W=68, 3-class B3_MTL; "
                     "one process, BLAS pinned to 1 thread" % (n1, seed, l_harm, l_perc),
           "cpu_model": cpu_model(), "host_cores_available": os.cpu_count(), "host_cores_usable": usable_cores()}
    nw = int(workers) if workers else min(usable_cores(), max_workers)
    if nw > 1:
        import multiprocessing as mp
        per_worker = int(max(4, min(1024, budget_s * single)))
        chunks = [(k * per_worker, (k + 1) * per_worker) for k in range(nw)]
        with mp.get_context("fork").Pool(nw, initializer=_init, initargs=(l_harm, l_perc, seed, model_seed)) as pool:
            pool.map(_chunk, [(0, 2)] * nw)  # every worker warm
            t0 = time.perf_counter()
            pool.map(_chunk, chunks, chunksize=1)
            dta = time.perf_counter() - t0
        res["all_cores"] = {"value": round(nw * per_worker / dta, 2), "unit": "clips/s", "cores": nw,
                            "sample": "%d clips: %d worker processes x %d clips each, 1 BLAS thread per worker"
                                      % (nw * per_worker, nw, per_worker)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--l-harm", type=int, default=17)
    ap.add_argument("--l-perc", type=int, default=17)
    ap.add_argument("--budget", type=float, default=10.0, help="seconds of CPU work per leg (1 core, all cores)")
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1000)
    a = ap.parse_args()
    print(json.dumps(measure(a.l_harm, a.l_perc, a.budget, a.workers or None, a.seed)))


if __name__ == "__main__":
    main()
