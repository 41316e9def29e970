"""Run only the HPSS median kernel (time-major harm, as in the fused pipeline) on the bench batch: target of the
rocprofv3 PMC passes of tools/gpu/pmc_median.sh.  Prints the event-timed mean."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.synth import synth_clips

B = int(os.environ.get("B", 1024))
lh, lp = int(os.environ.get("LH", 17)), int(os.environ.get("LP", 17))
iters = int(os.environ.get("ITERS", 20))
layout = int(os.environ.get("LAYOUT", 1))
fe = Frontend(FrontendConfig(l_harm=lh, l_perc=lp))
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1), (B // 64, 1))).cuda()
S = fe.stft_mag(audio)
T_ = S.shape[2]
harm = torch.empty((B, ((T_ + 15) // 16) * 16 * fe.K), device=S.device)  # large enough for every layout
perc = torch.empty_like(S)
p = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run = lambda: _lib.check(fe.lib.smh_hpss_median_ex_f32(fe._h, p(S), B, fe.K, S.shape[2], lh, lp, p(harm), p(perc), layout, st))
for _ in range(3):
    run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
print("layout=%d " % layout + "(%d,%d) B=%d: %.4f ms -> %.1f%% of 8 TB/s" % (lh, lp, B, ms, 100 * 3 * 201 * 98 * 4 * B / (ms * 1e-3) / 8e12), flush=True)

if layout == 2:  # decode (B, T/16, K, 16) and compare with the (B, K, T) result of the parity entry point
    G = (T_ + 15) // 16
    got = harm.view(B, G, fe.K, 16).permute(0, 2, 1, 3).reshape(B, fe.K, G * 16)[:, :, :T_]
    ref, _ = fe.hpss_median(S[:8], lh, lp)
    torch.cuda.synchronize()
    print("layout 2 decode equals (B,K,T):", bool(torch.equal(got[:8], ref)), flush=True)
