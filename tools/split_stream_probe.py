"""Throughput probe: 1024 clips per step as ONE batch on one stream, against the same clips as n part-batches (512 / 256 each) that go
down n streams at once -- part-grid kernels leave CUs free, so a memory-bound kernel of one part can run beside the issue-bound
network kernel of another (full-grid kernels of two streams do not share the chip: tools/two_stream_probe.py).  Timing only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
from sm_hpss_mtl_amd.synth import bench_clips

B = 1024
audio = torch.from_numpy(bench_clips(B, 0)).cuda()
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
full = HotPath(Frontend(FrontendConfig(l_harm=17, l_perc=17)), model, B, audio.shape[1])
for _ in range(100):
    full.step(audio)
torch.cuda.synchronize()


def run_full(K=300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        full.step(audio)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


def make(nparts):
    n = B // nparts
    paths = [HotPath(Frontend(FrontendConfig(l_harm=17, l_perc=17)), model, n, audio.shape[1]) for _ in range(nparts)]
    parts = [audio[i * n:(i + 1) * n].contiguous() for i in range(nparts)]
    streams = [torch.cuda.Stream() for _ in range(nparts)]
    return paths, parts, streams


def run_split(cfg, K=300, one_stream=False):
    paths, parts, streams = cfg
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for p, a, s in zip(paths, parts, streams):
            if one_stream:
                p.step(a)
            else:
                with torch.cuda.stream(s):
                    p.step(a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


c2, c4 = make(2), make(4)
for cfg in (c2, c4):
    for _ in range(20):
        run_split(cfg, K=2)
for rep in range(3):
    print("1024 as one batch %.4f ms | 2 x 512: one stream %.4f, two streams %.4f | 4 x 256: one stream %.4f, four streams %.4f" % (
        run_full(), run_split(c2, one_stream=True), run_split(c2), run_split(c4, one_stream=True), run_split(c4)), flush=True)
