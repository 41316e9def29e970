"""Do the STFT kernel (VALU-bound, 45 KB LDS) and the network kernel (MFMA-bound, 111 KB LDS, one workgroup per CU)
share the CUs when launched on two HIP streams?  Times both alone, back to back on one stream, and on two streams."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.synth import synth_clips

B = 1024
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1), (B // 64, 1))).cuda()
T = fe.num_frames(audio.shape[1])
S = torch.empty((B, fe.K, T), device="cuda")
x = torch.randn((B, 68, 240), device="cuda")
out = torch.empty((B, model.out_dim), device="cuda")
lib, h = fe.lib, fe._h
p = lambda t: C.c_void_p(t.data_ptr())

def stft(st):
    _lib.check(lib.smh_stft_mag_f32(h, p(audio), B, audio.shape[1], p(S), C.c_void_p(st.cuda_stream)))

def net(st):
    with torch.cuda.stream(st):
        model.forward_device(x, out=out)

def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

s0 = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
print("stft alone      %.4f ms" % timed(lambda: stft(s0)))
print("network alone   %.4f ms" % timed(lambda: net(s0)))
print("one stream      %.4f ms" % timed(lambda: (stft(s0), net(s0))))
def two():
    stft(sa); net(sb)
print("two streams     %.4f ms" % timed(two))
