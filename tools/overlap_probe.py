"""Does the model kernel (MFMA-bound) overlap with the next batch's front end (VALU/HBM-bound) on two HIP streams?
Steady-state clips/s of the bench pipeline: one stream vs front end on stream A + model on stream B."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.model import B3MTL
from sm_hpss_mtl_amd.synth import synth_clips

B, K_STEPS = 1024, 40
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
model = B3MTL(n_feat=240, patch_size=68, n_classes=3, seed=0)
audio = torch.from_numpy(np.tile(synth_clips(64, seed=1), (B // 64, 1))).cuda()
T = fe.num_frames(audio.shape[1])
dev = audio.device
S = torch.empty((B, fe.K, T), device=dev); harm = torch.empty_like(S); perc = torch.empty_like(S)
fv = torch.empty((B, 240, T), device=dev); mk = torch.empty(2 * B, dtype=torch.int32, device=dev)
patches = [torch.empty((B, 68, 240), device=dev) for _ in range(2)]
logits = [torch.empty((B, model.out_dim), device=dev) for _ in range(2)]
lib, h = fe.lib, fe._h
p = lambda t: C.c_void_p(t.data_ptr())

def front(st, pb):
    s = C.c_void_p(st.cuda_stream)
    _lib.check(lib.smh_stft_mag_f32(h, p(audio), B, audio.shape[1], p(S), s))
    lay = _lib.check(lib.smh_hpss_median_ex_f32(h, p(S), B, fe.K, T, 17, 17, p(harm), p(perc), 1, s))
    _lib.check(lib.smh_features_ex_f32(h, p(S), p(harm), p(perc), lay, B, T, 68, 68, p(fv), p(pb), p(mk), s))

def run(two_streams):
    sa = torch.cuda.Stream(); sb = torch.cuda.Stream() if two_streams else sa
    ev_front = [torch.cuda.Event() for _ in range(2)]; ev_model = [torch.cuda.Event() for _ in range(2)]
    model.forward_device(patches[0], out=logits[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K_STEPS):
        i = k & 1
        with torch.cuda.stream(sa):
            if k >= 2: sa.wait_event(ev_model[i])          # patches[i] free again
            front(sa, patches[i]); ev_front[i].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ev_front[i])
            model.forward_device(patches[i], out=logits[i]); ev_model[i].record(sb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * K_STEPS / dt, 1e3 * dt / K_STEPS

for mode in (False, True, False, True):
    cps, ms = run(mode)
    print("%s: %.0f clips/s, %.4f ms/step" % ("two streams" if mode else "one stream ", cps, ms), flush=True)
