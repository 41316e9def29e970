"""GPU timing of dense file-level inference (DAFx12_Speech_Music_Detection_B3_MTL_v2.py:594-706): one file's
featuregram (240, T), hop-1 patches of W frames, the 'M' head's probability per patch, then the 501-wide median.  Prints ms per
10 000-frame batch and patches per second, by parts."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import inference
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 68
rng = np.random.default_rng(0)
fv = torch.from_numpy(rng.standard_normal((240, T)).astype(np.float32) * 12 - 40).cuda()
model = B3MTL(n_feat=240, patch_size=W, n_classes=3, seed=0)
fe = Frontend(FrontendConfig())


def timed(fn, reps=10):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


ms, track = timed(lambda: inference.patch_probabilities(fv, model, W, 1, "M"))
n = len(track)
print("patch_probabilities (layer 0 once per frame, patches as windows): T = %d frames, W = %d -> %d patches: %.3f ms = %.2f M patches/s" % (T, W, n, ms, n / ms / 1e3), flush=True)
os.environ["SMH_DENSE_PATCHES"] = "1"
ms_b, track_b = timed(lambda: inference.patch_probabilities(fv, model, W, 1, "M"))
del os.environ["SMH_DENSE_PATCHES"]
print("patch_probabilities (patches built, SMH_DENSE_PATCHES=1): %.3f ms = %.2f M patches/s; max |difference| of the two tracks %.2e" % (
    ms_b, n / ms_b / 1e3, float(np.max(np.abs(track - track_b)))), flush=True)
std = torch.cat([fe.standardize_rows(fv[:120]), fe.standardize_rows(fv[120:])], dim=0)
ms_d, _ = timed(lambda: model.forward_dense(std, 1))
print("  forward_dense alone (l0_frames_kernel + the forward from windows): %.3f ms" % ms_d, flush=True)
R = 120
d = torch.cat([fe.standardize_rows(fv[:R]), fe.standardize_rows(fv[R:])], dim=0)


def gather():
    h = fe.extract_patches(fe.standardize_rows(d[:R])[None], W, 1, time_major=True)
    p = fe.extract_patches(fe.standardize_rows(d[R:])[None], W, 1, time_major=True)
    return torch.cat([h, p], dim=2)


ms_g, x = timed(gather)
ms_f, _ = timed(lambda: model.forward_device(x))
print("  parts: standardise + hop-1 patch gather + concat %.3f ms (%.0f MB of patches), forward from patches %.3f ms" % (ms_g, x.numel() * 4 / 1e6, ms_f), flush=True)
tr = torch.from_numpy(track).cuda()
ms_m, _ = timed(lambda: inference.medfilt(tr, 501))
print("  medfilt(501) over the %d-value track: %.3f ms" % (n, ms_m), flush=True)
