"""Device-resident batched front end: thin host wrapper over the C ABI (include/smh.h).

torch is plumbing here (device memory + the current HIP stream); every computation happens in
libsmh.so.  Inputs/outputs are float32 CUDA(=HIP) tensors, spectrogram-like tensors are (B, rows, T).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib

# featName -> (n_mels used?, log?)   (lib/preprocessing.py:404-444; only the '*HarmPerc*' branches are
# on the hot path -- SURVEY 8a)
FEATS = {
    "MelHarmPercSpec": (True, False),
    "LogMelHarmPercSpec": (True, True),
    "HarmPercSpec": (False, False),
    "LogHarmPercSpec": (False, True),
}


@dataclass(frozen=True)
class FrontendConfig:
    n_fft: int = 400
    win_length: int = 400
    hop: int = 160
    n_mels: int = 120
    l_harm: int = 21
    l_perc: int = 11
    log_db: bool = True
    mel_sr: float = 22050.0

    @staticmethod
    def from_params(PARAMS, n_fft, n_mels, featName, fs=16000):
        """Build from the reference's PARAMS dict (Proposed_Work_Results.py:723-807)."""
        if featName not in FEATS:
            raise ValueError("featName %r is not one of the HPSS feature names %s" % (featName, sorted(FEATS)))
        use_mel, log = FEATS[featName]
        model = PARAMS["Model"]
        return FrontendConfig(
            n_fft=int(n_fft), win_length=int(PARAMS["Tw"] * fs / 1000), hop=int(PARAMS["Ts"] * fs / 1000),
            n_mels=int(n_mels) if use_mel else 0, l_harm=int(PARAMS["l_harm"][model]),
            l_perc=int(PARAMS["l_perc"][model]), log_db=log)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return _lib.current_stream()


def _out(out, key, shape, dtype, dev):
    """A caller-supplied output tensor (steady-state loops allocate nothing) or a fresh one.  The C ABI receives raw
    pointers without sizes, so a stale `out` dict from a smaller batch must never reach it: wrong shape / dtype / device
    raises here."""
    t = out.get(key) if out else None
    if t is None:
        return torch.empty(shape, dtype=dtype, device=dev)
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("out[%r] must be a contiguous %s tensor on %s" % (key, dtype, dev))
    if tuple(t.shape) != tuple(shape):
        raise ValueError("out[%r] has shape %s, this call writes %s" % (key, tuple(t.shape), tuple(shape)))
    return t


def _f32c(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA/HIP torch tensor" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


class Frontend:
    """Owns one `smh_ctx` (window, FFT twiddles, mel CSR tables) for a fixed configuration."""

    def __init__(self, cfg: FrontendConfig = FrontendConfig()):
        self.lib = _lib.require_gpu()
        self.cfg = cfg
        c = _lib.FrontendCfg(cfg.n_fft, cfg.win_length, cfg.hop, cfg.n_mels, cfg.l_harm, cfg.l_perc,
                             1 if cfg.log_db else 0, cfg.mel_sr)
        h = C.c_void_p()
        _lib.check(self.lib.smh_ctx_create(C.byref(c), C.byref(h)), "smh_ctx_create")
        self._h = h
        self.K = 1 + cfg.n_fft // 2
        self.rows = self.lib.smh_ctx_feat_rows(self._h)
        self._work = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.smh_ctx_destroy(h)
            self._h = None

    # ---- integer contracts ----
    def num_frames(self, n_samples: int) -> int:
        return self.lib.smh_num_frames(n_samples, self.cfg.n_fft, self.cfg.hop)

    def num_patches(self, T: int, W: int, shift: int) -> int:
        return _lib.check(self.lib.smh_num_patches(self.lib.smh_tiled_frames(T, W), W, shift), "smh_num_patches")

    def mel_basis(self) -> np.ndarray:
        out = np.empty((self.cfg.n_mels, self.K), np.float32)
        _lib.check(self.lib.smh_ctx_mel_basis(self._h, out.ctypes.data_as(C.c_void_p)), "smh_ctx_mel_basis")
        return out

    # ---- stage-by-stage API (one call per reference call) ----
    def stft_mag(self, audio):
        audio = _f32c(audio, "audio")
        B, N = audio.shape
        T = self.num_frames(N)
        if T < 1:
            raise ValueError("clip of %d samples is shorter than n_fft=%d" % (N, self.cfg.n_fft))
        S = torch.empty((B, self.K, T), dtype=torch.float32, device=audio.device)
        _lib.check(self.lib.smh_stft_mag_f32(self._h, _ptr(audio), B, N, _ptr(S), _stream()), "smh_stft_mag_f32")
        return S

    def hpss_median(self, S, l_harm=None, l_perc=None):
        S = _f32c(S, "S")
        B, K, T = S.shape
        lh = self.cfg.l_harm if l_harm is None else l_harm
        lp = self.cfg.l_perc if l_perc is None else l_perc
        harm, perc = torch.empty_like(S), torch.empty_like(S)
        _lib.check(self.lib.smh_hpss_median_f32(self._h, _ptr(S), B, K, T, lh, lp, _ptr(harm), _ptr(perc), _stream()),
                   "smh_hpss_median_f32")
        return harm, perc

    def median_time(self, S, l_harm):
        S = _f32c(S, "S")
        B, K, T = S.shape
        out = torch.empty_like(S)
        _lib.check(self.lib.smh_median_time_f32(self._h, _ptr(S), B, K, T, l_harm, _ptr(out), _stream()),
                   "smh_median_time_f32")
        return out

    def median_freq(self, S, l_perc):
        S = _f32c(S, "S")
        B, K, T = S.shape
        out = torch.empty_like(S)
        _lib.check(self.lib.smh_median_freq_f32(self._h, _ptr(S), B, K, T, l_perc, _ptr(out), _stream()),
                   "smh_median_freq_f32")
        return out

    def softmask(self, S, harm, perc):
        S, harm, perc = _f32c(S, "S"), _f32c(harm, "harm"), _f32c(perc, "perc")
        if not (S.shape == harm.shape == perc.shape):
            raise ValueError("softmask: shape mismatch %s %s %s" % (S.shape, harm.shape, perc.shape))
        H, P = torch.empty_like(S), torch.empty_like(S)
        _lib.check(self.lib.smh_softmask_f32(self._h, _ptr(S), _ptr(harm), _ptr(perc), S.numel(), _ptr(H), _ptr(P),
                                             _stream()), "smh_softmask_f32")
        return H, P

    def mel(self, X):
        X = _f32c(X, "X")
        B, K, T = X.shape
        if K != self.K:
            raise ValueError("mel: expected %d bins, got %d" % (self.K, K))
        Y = torch.empty((B, self.cfg.n_mels, T), dtype=torch.float32, device=X.device)
        _lib.check(self.lib.smh_mel_f32(self._h, _ptr(X), B, T, _ptr(Y), _stream()), "smh_mel_f32")
        return Y

    def power_to_db_sq(self, X):
        """power_to_db(X**2) with the top_db max taken per leading-axis entry."""
        X = _f32c(X, "X")
        n = X.shape[0]
        Y = torch.empty_like(X)
        _lib.check(self.lib.smh_power_to_db_sq_f32(self._h, _ptr(X), n, X.numel() // max(n, 1), _ptr(Y), _stream()),
                   "smh_power_to_db_sq_f32")
        return Y

    def standardize_rows(self, X):
        X = _f32c(X, "X")
        T = X.shape[-1]
        Y = torch.empty_like(X)
        _lib.check(self.lib.smh_standardize_rows_f32(self._h, _ptr(X), X.numel() // T, T, _ptr(Y), _stream()),
                   "smh_standardize_rows_f32")
        return Y

    def extract_patches(self, FV, W, shift, time_major=False):
        FV = _f32c(FV, "FV")
        B, F, T = FV.shape
        nP = self.num_patches(T, W, shift)
        shape = (B * nP, W, F) if time_major else (B * nP, F, W)
        out = torch.empty(shape, dtype=torch.float32, device=FV.device)
        got = _lib.check(self.lib.smh_extract_patches_f32(self._h, _ptr(FV), B, F, T, W, shift, 1 if time_major else 0,
                                                          _ptr(out) if nP else None, _stream()), "smh_extract_patches_f32")
        assert got == nP
        return out

    def features(self, S, harm, perc, W=None, shift=None, out=None):
        """(S, harm, perc) -> dict(fv[, patches]): masks + mel + dB, then standardise + time-major patches."""
        S, harm, perc = _f32c(S, "S"), _f32c(harm, "harm"), _f32c(perc, "perc")
        B, K, T = S.shape
        dev = S.device
        fv = _out(out, "fv", (B, 2 * self.rows, T), torch.float32, dev)
        nP, patches = 0, None
        if W is not None:
            nP = self.num_patches(T, W, shift)
            patches = _out(out, "patches", (B * nP, W, 2 * self.rows), torch.float32, dev)
        keys = (out or {}).get("maxkeys")
        if keys is None:
            keys = torch.empty(2 * max(B, 1), dtype=torch.int32, device=dev)
        elif not (keys.is_cuda and keys.dtype == torch.int32 and keys.numel() >= 2 * B and keys.is_contiguous()):
            raise ValueError("out['maxkeys'] must be a contiguous int32 device tensor with at least 2*B = %d entries" % (2 * B))
        got = _lib.check(self.lib.smh_features_f32(self._h, _ptr(S), _ptr(harm), _ptr(perc), B, T, W or 0, shift or 0,
                                                   _ptr(fv), _ptr(patches) if nP else None, _ptr(keys), _stream()),
                         "smh_features_f32")
        assert got == nP
        return {"fv": fv, "patches": patches, "n_patches": nP, "maxkeys": keys}

    def features_l0(self, S, harm, perc, harm_layout, W, shift, model, out=None, patches=False):
        """`features` fused with the first layer of `model` (B3MTL): returns dict(fv, x0p (B*nP, 2, W, 32)[, patches]).
        Feed x0p to `model.forward_from_x0`.  Same logits as features -> forward_device within f32 tolerance."""
        S, harm, perc = _f32c(S, "S"), _f32c(harm, "harm"), _f32c(perc, "perc")
        B, K, T = S.shape
        if int(harm_layout) == 2 and harm.numel() < B * self.lib.smh_harm_buffer_floats(K, T):
            raise ValueError("harm_layout 2 needs smh_harm_buffer_floats(K, T) floats per clip")
        if model.n_feat != 2 * self.rows or model.patch_size != W:
            raise ValueError("model expects (W=%d, n_feat=%d), the front end produces (W=%d, n_feat=%d)"
                             % (model.patch_size, model.n_feat, W, 2 * self.rows))
        model._sync_weights()
        nP = self.num_patches(T, W, shift)
        dev = S.device
        fv = _out(out, "fv", (B, 2 * self.rows, T), torch.float32, dev)
        x0p = _out(out, "x0p", (B * nP, 2, W, 32), torch.float32, dev)
        pt = _out(out, "patches", (B * nP, W, 2 * self.rows), torch.float32, dev) if patches else None
        keys = (out or {}).get("maxkeys")
        if keys is None:
            keys = torch.empty(2 * max(B, 1), dtype=torch.int32, device=dev)
        elif not (keys.is_cuda and keys.dtype == torch.int32 and keys.numel() >= 2 * B and keys.is_contiguous()):
            raise ValueError("out['maxkeys'] must be a contiguous int32 device tensor with at least 2*B = %d entries" % (2 * B))
        got = _lib.check(self.lib.smh_features_l0_f32(
            self._h, _ptr(S), _ptr(harm), _ptr(perc), int(harm_layout), B, T, W, shift, _ptr(fv), _ptr(pt),
            C.c_void_p(self.lib.smh_model_w0_ptr(model._h)), _ptr(x0p), _ptr(keys), _stream()), "smh_features_l0_f32")
        assert got == nP
        return {"fv": fv, "x0p": x0p, "patches": pt, "n_patches": nP, "maxkeys": keys}

    # ---- fused fast path ----
    def run(self, audio, W=None, shift=None, taps=False, out=None):
        """audio (B, n_samples) -> dict(fv=(B, 2*rows, T)[, patches=(B*nP, W, 2*rows)][, S, harm, perc]).
        `out` may carry preallocated 'fv' / 'patches' tensors (steady-state loops allocate nothing)."""
        audio = _f32c(audio, "audio")
        B, N = audio.shape
        T = self.num_frames(N)
        if T < 1:
            raise ValueError("clip of %d samples is shorter than n_fft=%d" % (N, self.cfg.n_fft))
        dev = audio.device
        out = {} if out is None else out
        fv = out.get("fv")
        if fv is None or fv.shape != (B, 2 * self.rows, T):
            fv = torch.empty((B, 2 * self.rows, T), dtype=torch.float32, device=dev)
        patches, nP = None, 0
        if W is not None:
            nP = self.num_patches(T, W, shift)
            patches = out.get("patches")
            if patches is None or patches.shape != (B * nP, W, 2 * self.rows):
                patches = torch.empty((B * nP, W, 2 * self.rows), dtype=torch.float32, device=dev)
        need = self.lib.smh_frontend_workspace_bytes(self._h, B, N)
        if self._work is None or self._work.numel() < need or self._work.device != dev:
            self._work = torch.empty(need, dtype=torch.uint8, device=dev)
        S = harm = perc = None
        if taps:
            S = torch.empty((B, self.K, T), dtype=torch.float32, device=dev)
            harm, perc = torch.empty_like(S), torch.empty_like(S)
        got = _lib.check(self.lib.smh_frontend_f32(
            self._h, _ptr(audio), B, N, W or 0, shift or 0, _ptr(fv), _ptr(patches) if nP else None,
            _ptr(self._work), self._work.numel(), _ptr(S), _ptr(harm), _ptr(perc), _stream()), "smh_frontend_f32")
        assert got == nP, (got, nP)
        res = {"fv": fv, "n_patches": nP}
        if W is not None:
            res["patches"] = patches
        if taps:
            res.update(S=S, harm=harm, perc=perc)
        return res

    # ---- ragged batches ----
    def run_ragged(self, clips, W=None, shift=None):
        """Clips of DIFFERENT lengths in one call (`smh_frontend_ragged_f32`).  clips: list of 1-D float32 arrays / tensors.
        Returns dict(fv=[(2*rows, T_b) tensors], patches=[(nP_b, W, 2*rows) tensors] (views of one buffer each),
        n_patches=[...], T=[...]).  Every clip gets bit for bit what `run` gives it alone or in an equal-length batch:
        the clips are laid out at 16-byte aligned offsets, so each takes the same kernels as there."""
        B = len(clips)
        if B == 0:
            return {"fv": [], "patches": [], "n_patches": [], "T": []}
        dev = torch.device("cuda", torch.cuda.current_device())
        lens = [int(c.shape[0]) for c in clips]
        offs, o = [], 0
        for n in lens:
            offs.append(o)
            o += (n + 3) // 4 * 4  # next clip starts on a 16-byte boundary
        for c in clips:
            if c.ndim != 1:
                raise ValueError("run_ragged: every clip must be 1-D")
        if all(isinstance(c, np.ndarray) for c in clips):
            # host clips: laid out in ONE host buffer and uploaded by ONE copy (a copy per clip is a host synchronisation per clip)
            host = np.zeros(max(o, 1), dtype=np.float32)
            for c, n, of in zip(clips, lens, offs):
                host[of:of + n] = c
            audio = torch.from_numpy(host).to(device=dev)
        else:
            audio = torch.zeros(max(o, 1), dtype=torch.float32, device=dev)
            for c, n, of in zip(clips, lens, offs):
                t = c if isinstance(c, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(c, dtype=np.float32))
                audio[of:of + n] = t.to(device=dev, dtype=torch.float32)
        h_off = (C.c_longlong * B)(*offs)
        h_len = (C.c_int * B)(*lens)
        fv_off, p_off = (C.c_longlong * (B + 1))(), (C.c_longlong * (B + 1))()
        hT, hnP = (C.c_int * B)(), (C.c_int * B)()
        work = C.c_size_t()
        _lib.check(self.lib.smh_frontend_ragged_sizes(self._h, h_off, h_len, B, W or 0, shift or 0, fv_off, p_off, hT, hnP,
                                                      C.byref(work)), "smh_frontend_ragged_sizes")
        F = 2 * self.rows
        fv = torch.empty(max(int(fv_off[B]), 1), dtype=torch.float32, device=dev)
        patches = torch.empty((max(int(p_off[B]), 1), W or 1, F), dtype=torch.float32, device=dev) if W else None
        if self._work is None or self._work.numel() < work.value or self._work.device != dev:
            self._work = torch.empty(max(work.value, 1), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.smh_frontend_ragged_f32(self._h, _ptr(audio), h_off, h_len, B, W or 0, shift or 0, _ptr(fv),
                                                    _ptr(patches) if (W and int(p_off[B]) > 0) else None, _ptr(self._work),
                                                    self._work.numel(), _stream()), "smh_frontend_ragged_f32")
        res = {"fv": [fv[int(fv_off[b]):int(fv_off[b + 1])].view(F, int(hT[b])) for b in range(B)],
               "T": [int(hT[b]) for b in range(B)], "n_patches": [int(hnP[b]) for b in range(B)]}
        if W:
            res["patches"] = [patches[int(p_off[b]):int(p_off[b + 1])] for b in range(B)]
        return res

    def patches_from_featuregram(self, fv, W, shift):
        """get_feature_patches for the TCN models on the device: featuregram (2*rows, T) [rows 0..F/2-1 harmonic] ->
        standardised time-major patches (nP, W, 2*rows) (tile-if-short, StandardScaler per half, extract_patches, transpose)."""
        fv = _f32c(fv, "fv")
        if fv.dim() != 2:
            raise ValueError("FV should be of the shape (nFeatures, nFrames)")
        x = self.standardize_rows(fv)  # per row over the frames: the per-half scaler is row-wise, so halves need no split
        return self.extract_patches(x[None], W, shift, time_major=True)
