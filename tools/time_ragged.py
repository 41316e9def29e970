"""GPU timing of the ragged front end (smh_frontend_ragged_f32: files of DIFFERENT lengths in one call, what the reference's generators
feed it one file at a time, Proposed_Work_Results.py:92-95, 465-474): B files of 1..MAXS seconds, featuregram + W = 68 / shift 68
patches.  Reports files/s and seconds of audio per second, against the same amount of audio as equal-length 1 s clips in one batch."""
import os, sys, time
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig, _ptr, _stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MAXS = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rng = np.random.default_rng(0)
lens = [int(rng.uniform(1.0, MAXS) * 16000) // 2 * 2 for _ in range(B)]
if os.environ.get("RAG_T_PARITY"):  # tuning: every file an even (0) or odd (1) number of frames
    par = int(os.environ["RAG_T_PARITY"])
    lens = [n if (1 + (n - 400) // 160) % 2 == par else n + 160 for n in lens]
fe = Frontend(FrontendConfig(l_harm=21, l_perc=11))
offs, o = [], 0
for n in lens:
    offs.append(o)
    o += (n + 3) // 4 * 4
audio = (torch.rand(o, device="cuda") - 0.5)
h_off, h_len = (C.c_longlong * B)(*offs), (C.c_int * B)(*lens)
fv_off, p_off = (C.c_longlong * (B + 1))(), (C.c_longlong * (B + 1))()
hT, hnP = (C.c_int * B)(), (C.c_int * B)()
work = C.c_size_t()
_lib.check(fe.lib.smh_frontend_ragged_sizes(fe._h, h_off, h_len, B, 68, 68, fv_off, p_off, hT, hnP, C.byref(work)))
fv = torch.empty(int(fv_off[B]), device="cuda")
patches = torch.empty((int(p_off[B]), 68, 240), device="cuda")
wk = torch.empty(work.value, dtype=torch.uint8, device="cuda")
run = lambda: _lib.check(fe.lib.smh_frontend_ragged_f32(fe._h, _ptr(audio), h_off, h_len, B, 68, 68, _ptr(fv), _ptr(patches), _ptr(wk), wk.numel(), _stream()))
for _ in range(3):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    run()
host_ms = (time.perf_counter() - t0) / K * 1e3  # the calls have returned, the device may still be busy
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print("host time to enqueue one call: %.2f ms (%.1f us per file)" % (host_ms, host_ms / B * 1e3), flush=True)
secs = sum(lens) / 16000.0
print("ragged: %d files of 1-%.0f s (%.0f s of audio, %d frames, %d patches): %.2f ms per call = %.0f files/s = %.0f s of audio per second" % (
    B, MAXS, secs, sum(hT), int(p_off[B]), ms, B / ms * 1e3, secs / ms * 1e3), flush=True)
# the same amount of audio as one batch of 1 s clips
Be = int(secs)
a1 = (torch.rand((Be, 16000), device="cuda") - 0.5)
out = fe.run(a1, W=68, shift=68)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    out = fe.run(a1, W=68, shift=68)
torch.cuda.synchronize()
ms1 = (time.perf_counter() - t0) / K * 1e3
print("equal-length batch of %d one-second clips: %.2f ms per call = %.0f s of audio per second" % (Be, ms1, Be / ms1 * 1e3), flush=True)
