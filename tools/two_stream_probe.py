"""Throughput probe: the bench path on ONE stream (batch after batch) against TWO streams that take alternate batches (each
with its own front-end context and buffers; one model, read-only).  Timing only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
from sm_hpss_mtl_amd.synth import synth_clips

B = 1024
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1), (B // 64, 1))).cuda()
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
paths = [HotPath(Frontend(FrontendConfig(l_harm=17, l_perc=17)), model, B, audio.shape[1]) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for _ in range(100):
    paths[0].step(audio)
torch.cuda.synchronize()

def run(nstreams, K=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        i = k % nstreams
        with torch.cuda.stream(streams[i]):
            paths[i].step(audio)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

for rep in range(3):
    print("one stream %.4f ms per batch   two streams %.4f ms per batch" % (run(1), run(2)), flush=True)
