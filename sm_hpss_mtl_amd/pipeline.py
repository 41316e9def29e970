"""The hot path as ONE object with preallocated device buffers: audio (B, n_samples) -> logits (B*nP, out_dim).

    STFT -> HPSS medians (l_harm x l_perc) -> soft masks -> mel -> dB -> standardise -> patches -> B3_MTL forward

= `get_featuregram` (lib/preprocessing.py:414-424) + `get_feature_patches` (:137-142, 208-234) + `model.predict`
(Proposed_Work_Results.py:459-496, 520) for a batch of equal-length clips that is already resident in HBM.  Four
launches per step through the C ABI (include/smh.h), no allocation, no host synchronisation.

`bench.py` times exactly `HotPath.step`; `tests/test_bench_path_gpu.py` compares exactly `HotPath.step` with the
oracle -- the timed configuration and the tested configuration are the same code.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

STAGES = ("stft", "median", "features", "model")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class HotPath:
    def __init__(self, fe, model, batch, n_samples, patch=68, shift=None, fuse_l0=True, two_kernel_features=False,
                 model_dtype="f32", keep_patches=False, keep_trunk=False, device=None):
        """fe: Frontend, model: B3MTL, or None for the front end alone (BASELINE config 2, "HPSS-only": three launches, the step
        ends with the featuregram; no patches, no logits).  fuse_l0: the network's first 1x1 convolution runs inside the feature kernel
        (smh_features_l0_f32 + smh_model_forward_x0_f32) instead of patches -> smh_model_forward_f32.
        keep_patches / keep_trunk: also write the standardised time-major patches / the TCN output (N, W, 32) -- parity taps
        that `model.predict` does not return; never set by bench.py."""
        self.fe, self.model, self.lib, self._h = fe, model, fe.lib, fe._h
        self.B, self.n_samples, self.W = int(batch), int(n_samples), int(patch)
        self.shift = int(patch if shift is None else shift)
        self.model_dtype = model_dtype
        if model_dtype not in ("f32", "bf16"):
            raise ValueError("model_dtype must be 'f32' or 'bf16'")
        self.fuse_l0 = bool(fuse_l0) and model is not None  # both network dtypes start from the layer-0 partials
        self.T = fe.num_frames(self.n_samples)
        if self.T < 1:
            raise ValueError("clip of %d samples is shorter than n_fft=%d" % (n_samples, fe.cfg.n_fft))
        if model is not None and (model.n_feat != 2 * fe.rows or model.patch_size != self.W):
            raise ValueError("model expects (W=%d, n_feat=%d), the front end produces (W=%d, n_feat=%d)"
                             % (model.patch_size, model.n_feat, self.W, 2 * fe.rows))
        self.nP = fe.num_patches(self.T, self.W, self.shift) if model is not None else 0
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        B, K, T, F = self.B, fe.K, self.T, 2 * fe.rows
        f32 = dict(dtype=torch.float32, device=dev)
        self.S = torch.empty((B, K, T), **f32)
        self.perc = torch.empty_like(self.S)
        self.harm = torch.empty((B, self.lib.smh_harm_buffer_floats(K, T)), **f32)  # room for every harm layout
        self.fv = torch.empty((B, F, T), **f32)
        self.maxkeys = torch.empty(2 * max(B, 1), dtype=torch.int32, device=dev)
        need_patches = (keep_patches or not self.fuse_l0) and model is not None
        self.patches = torch.empty((B * self.nP, self.W, F), **f32) if need_patches else None
        self.x0p = torch.empty((B * self.nP, 2, self.W, 32), **f32) if self.fuse_l0 else None
        self.logits = torch.empty((B * self.nP, model.out_dim), **f32) if model is not None else None
        self.trunk = torch.empty((B * self.nP, self.W, 32), **f32) if (keep_trunk and model_dtype == "f32" and model is not None) else None
        # harmonic median layout: 16-frame blocks when the single feature kernel takes the clip, else time-major
        blocked = self.lib.smh_features_blocked_ok(self._h, T, 1 if self.fuse_l0 else 0) and not two_kernel_features
        self.want_layout = 2 if blocked else 1
        self.layout = None  # what the median launch actually wrote (set by step)

    def step(self, audio, record=None):
        """One pass over one batch.  `record`: five torch.cuda.Event (timing) recorded on the launch stream around
        the four stages, or None.  Returns the logits tensor (B*nP, out_dim) = [S | M | (N) | R | 3C]."""
        if audio.shape != (self.B, self.n_samples) or audio.dtype != torch.float32 or not audio.is_cuda:
            raise ValueError("audio must be a float32 device tensor of shape (%d, %d)" % (self.B, self.n_samples))
        lib, h, fe, m = self.lib, self._h, self.fe, self.model
        st = _lib.current_stream()
        if m is not None:
            m._sync_weights()
        if record is not None:
            record[0].record()
        _lib.check(lib.smh_stft_mag_f32(h, _p(audio), self.B, self.n_samples, _p(self.S), st), "smh_stft_mag_f32")
        if record is not None:
            record[1].record()
        lay = _lib.check(lib.smh_hpss_median_ex_f32(h, _p(self.S), self.B, fe.K, self.T, fe.cfg.l_harm, fe.cfg.l_perc,
                                                    _p(self.harm), _p(self.perc), self.want_layout, st),
                         "smh_hpss_median_ex_f32")
        self.layout = lay
        if record is not None:
            record[2].record()
        if self.fuse_l0:
            got = _lib.check(lib.smh_features_l0_f32(h, _p(self.S), _p(self.harm), _p(self.perc), lay, self.B, self.T,
                                                     self.W, self.shift, _p(self.fv), _p(self.patches),
                                                     C.c_void_p(lib.smh_model_w0_ptr(m._h)), _p(self.x0p),
                                                     _p(self.maxkeys), st), "smh_features_l0_f32")
        else:
            got = _lib.check(lib.smh_features_ex_f32(h, _p(self.S), _p(self.harm), _p(self.perc), lay, self.B, self.T,
                                                     self.W if m is not None else 0, self.shift if m is not None else 0,
                                                     _p(self.fv), _p(self.patches), _p(self.maxkeys), st), "smh_features_ex_f32")
        if got != self.nP:
            raise RuntimeError("feature stage produced %d patches per clip, expected %d" % (got, self.nP))
        if record is not None:
            record[3].record()
        if m is None:
            if record is not None:
                record[4].record()
            return self.fv
        if self.fuse_l0:
            m.forward_from_x0(self.x0p, out=self.logits, trunk=self.trunk, dtype=self.model_dtype)
        else:
            m.forward_device(self.patches, out=self.logits, trunk=self.trunk, dtype=self.model_dtype)
        if record is not None:
            record[4].record()
        return self.logits

    def harm_bkt(self):
        """The harmonic medians of the last step decoded to the reference's (B, K, T) layout (parity tap)."""
        B, K, T = self.B, self.fe.K, self.T
        if self.layout == 2:
            G = (T + 15) // 16
            return self.harm[:, :G * K * 16].view(B, G, K, 16).permute(0, 2, 1, 3).reshape(B, K, G * 16)[:, :, :T]
        if self.layout == 1:
            return self.harm[:, :K * T].view(B, T, K).transpose(1, 2)
        return self.harm[:, :K * T].view(B, K, T)
