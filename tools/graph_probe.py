"""Small-batch latency of the hot path: eager launches (four C-ABI calls per step from Python) against one hipGraph replay of the
same four launches (torch.cuda.CUDAGraph capture of HotPath.step).  Prints host-inclusive time per step, back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
from sm_hpss_mtl_amd.synth import bench_clips

model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
for B in (8, 32, 128, 1024):
    fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
    hp = HotPath(fe, model, B, 16000)
    audio = torch.from_numpy(bench_clips(B, 0)).cuda()
    for _ in range(20):
        hp.step(audio)
    torch.cuda.synchronize()
    ref = hp.logits.clone()

    def timed(fn, K=300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e6

    eager = timed(lambda: hp.step(audio))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            hp.step(audio)
            with torch.cuda.graph(g, stream=s):
                hp.step(audio)
        torch.cuda.current_stream().wait_stream(s)
        hp.logits.zero_()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(hp.logits, ref)
        graph = timed(g.replay)
        print("B = %4d: eager %.1f us per step, graph replay %.1f us per step, logits identical: %s" % (B, eager, graph, same), flush=True)
    except Exception as e:  # noqa
        print("B = %4d: eager %.1f us per step; capture failed: %s" % (B, eager, str(e)[:200]), flush=True)
