"""GPU timing of the STFT kernel alone on the bench batch (1024 distinct clips), per SMH_STFT_FRAMES setting (frames per workgroup,
threads).  Back-to-back launches: steady-state kernel time incl. the overlap of consecutive launches' tails."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sm_hpss_mtl_amd import _lib
from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
from sm_hpss_mtl_amd.synth import bench_clips
B = 1024
fe = Frontend(FrontendConfig(l_harm=17, l_perc=17))
audio = torch.from_numpy(bench_clips(B, 0)).cuda()
S = torch.empty((B, 201, 98), device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
run = lambda: _lib.check(fe.lib.smh_stft_mag_f32(fe._h, C.c_void_p(audio.data_ptr()), B, 16000, C.c_void_p(S.data_ptr()), st))
for cfg in sys.argv[1:] or ["20,256"]:
    parts = cfg.split(",")
    os.environ["SMH_STFT_FRAMES"] = ",".join(parts[:2])
    if len(parts) > 2:  # third field: table row pitch 25 | 40 (probe SMH_STFT_ROW, needs SMH_ENABLE_PROBES=1)
        os.environ["SMH_STFT_ROW"] = parts[2]
    for _ in range(30): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): run()
    e1.record(); torch.cuda.synchronize()
    print("frames,threads = %-8s %.1f us" % (cfg, e0.elapsed_time(e1) / 300 * 1e3))
