"""What fusing the median kernel behind the STFT could save at most: stage times of the bench path with
SMH_STFT_PROBE_NOSTORE (S computed, not stored) and SMH_MEDIAN_PROBE_NOLOAD (medians on a tile that is already in LDS).
Timing experiment: with a probe set the stage outputs are not results."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
B, NS = 1024, 16000
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
m = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
hp = HotPath(fe, m, B, NS, patch=68)
audio = torch.randn((B, NS), device="cuda") * 0.1
for _ in range(5): hp.step(audio)
torch.cuda.synchronize()
acc = np.zeros(4); n = 30
for _ in range(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    hp.step(audio, record=ev); torch.cuda.synchronize()
    acc += [ev[i].elapsed_time(ev[i + 1]) * 1000 for i in range(4)]
acc /= n
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(50): hp.step(audio)
t1.record(); torch.cuda.synchronize()
print("probes: stft_nostore=%s median_noload=%s | stft %.1f  median %.1f  features %.1f  model %.1f us | step %.1f us" % (
    bool(os.environ.get("SMH_STFT_PROBE_NOSTORE")), bool(os.environ.get("SMH_MEDIAN_PROBE_NOLOAD")), *acc, t0.elapsed_time(t1) * 1000 / 50))
