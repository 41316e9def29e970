"""How the step time settles on a device that was idle (the driver's 5 + 20 protocol lands inside this ramp): the bench's HotPath,
every one of the first STEPS steps timed with its own event pair, after IDLE_S seconds of an idle GPU.  Prints the per-step times."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.pipeline import HotPath
from sm_hpss_mtl_amd.synth import bench_clips
STEPS = int(os.environ.get("STEPS", "120"))
IDLE_S = float(os.environ.get("IDLE_S", "5"))
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
audio = torch.from_numpy(bench_clips(1024, 0)).cuda()
hp = HotPath(fe, model, 1024, audio.shape[1], patch=68, fuse_l0=True)
for rep in range(2):
    torch.cuda.synchronize()
    time.sleep(IDLE_S)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)]
    ev[0].record()
    cpu = []
    for k in range(STEPS):
        c0 = time.perf_counter()
        hp.step(audio, None)
        ev[k + 1].record()
        cpu.append(1e3 * (time.perf_counter() - c0))
    torch.cuda.synchronize()
    t = [ev[k].elapsed_time(ev[k + 1]) for k in range(STEPS)]
    print("  host time per step (ms), every 5th: " + " ".join("%.3f" % v for v in cpu[::5]))
    print("after %.0f s idle: steps 1-5 %s | 6-25 mean %.4f | 26-50 mean %.4f | 51-100 mean %.4f | last 20 mean %.4f ms" % (
        IDLE_S, " ".join("%.3f" % v for v in t[:5]), sum(t[5:25]) / 20, sum(t[25:50]) / 25, sum(t[50:100]) / 50, sum(t[-20:]) / 20))
    print("  every 5th step: " + " ".join("%.3f" % v for v in t[::5]))
