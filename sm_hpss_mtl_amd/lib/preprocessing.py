"""Counterpart of lib/preprocessing.py for the hot path -- same function names, argument meaning and
return types as the reference, computed by the HIP library.

  get_featuregram      (preprocessing.py:355-457)  '*HarmPerc*' branches 404-444
  get_feature_patches  (preprocessing.py:137-292)
  normalize_signal     (preprocessing.py:114-132)
  mix_signals          (preprocessing.py:297-325)
plus the batched, device-resident fast path the reference does not have:
  featuregram_batch / feature_patches_batch.

Per-file functions are thin wrappers: one clip is a batch of one.  `load_and_preprocess_signal` (SURVEY 8f
rank 1) decodes .wav/.npy on the host and runs normalise -> rms -> removeSilence -> normalise on the device
(`smh_preprocess_signal_f32`); `preprocess_signal_batch` is the device-resident form for equal-length clips.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import frontend as _fe

_frontends = {}


def _frontend_for(cfg: _fe.FrontendConfig) -> _fe.Frontend:
    fe = _frontends.get(cfg)
    if fe is None:
        fe = _frontends[cfg] = _fe.Frontend(cfg)
    return fe


# ---- signal conditioning (host, numpy: a few vector ops per file) -------------------------------------
def normalize_signal(Xin):
    """preprocessing.py:130-131"""
    Xin = Xin - np.mean(Xin)
    Xin = Xin / np.max(np.abs(Xin))
    return Xin


def mix_signals(Xin_sp, Xin_mu, target_dB):
    """preprocessing.py:297-325 (music looped to the speech length, scaled to the target SMR)."""
    sig_sp_len = len(Xin_sp)
    Xin_mu_temp = Xin_mu.copy()
    while len(Xin_mu_temp) < sig_sp_len:
        Xin_mu_temp = np.append(Xin_mu_temp, Xin_mu)
    common_len = min(sig_sp_len, len(Xin_mu_temp))
    Xin_sp = Xin_sp[:common_len]
    Xin_mu = Xin_mu_temp[:common_len]
    sig_sp_energy = np.sum(np.power(Xin_sp, 2)) / len(Xin_sp)
    sig_mu_energy = np.sum(np.power(Xin_mu, 2)) / len(Xin_mu)
    req_sig_mu_energy = sig_sp_energy / np.power(10, (target_dB / 10))
    sig_mu_mult_fact = np.sqrt(req_sig_mu_energy / sig_mu_energy)
    sig_sp_mult_fact = 1
    mult_fact_sum = sig_mu_mult_fact + sig_sp_mult_fact
    sig_mu_mult_fact /= mult_fact_sum
    sig_sp_mult_fact /= mult_fact_sum
    dt = Xin_sp.dtype
    Xin_mix = (dt.type(sig_sp_mult_fact) * Xin_sp + dt.type(sig_mu_mult_fact) * Xin_mu).astype(dt)
    return normalize_signal(Xin_mix)


def removeSilence(Xin, fs, Tw, Ts, alpha=0.025, beta=0.075):
    """preprocessing.py:21-110, the pure-Python sibling of tools.removeSilence (not on the reference's own call
    path: load_and_preprocess_signal uses the Cython one).  Same frame / sample markers (computed on the device;
    the threshold is the Cython function's float32 one), but the output is really shortened: the samples before the
    first removed run and between runs -- like the reference, the piece after the LAST run is dropped -- and one run
    is enough to trigger it.  Returns (Xin_silrem, sample_silMarker float64, frame_silMarker int, totalSilDuration)."""
    from .. import silence as _sil
    Xin = np.asarray(Xin)
    frameSize, frameShift = int((Tw * fs) / 1000), int((Ts * fs) / 1000)
    d = torch.from_numpy(np.ascontiguousarray(Xin, dtype=np.float32)).cuda()
    energy = _sil.rms(d, frameSize, frameShift)
    _, _, sm, fm = _sil.remove_silence(d, energy, fs, Tw, Ts, alpha, beta, markers=True)
    sample_silMarker = sm[0].cpu().numpy().astype(np.float64)
    frame_silMarker = fm[0].cpu().numpy().astype(int)
    dd = np.diff(np.concatenate([[1.0], sample_silMarker, [1.0]]))
    starts, ends = np.where(dd == -1)[0], np.where(dd == 1)[0]
    totalSilDuration = float(sum((l - k) / fs for k, l in zip(starts, ends)))
    if len(starts) > 0:
        pieces = [Xin[:starts[0]]] + [Xin[ends[i - 1]:starts[i]] for i in range(1, len(starts))]
        Xin_silrem = np.concatenate(pieces) if len(pieces) > 1 else pieces[0]
    else:
        Xin_silrem = Xin
    return Xin_silrem, sample_silMarker, frame_silMarker, totalSilDuration


def mix_signals_batch(Xin_sp, Xin_mu, target_dB):
    """Device-resident form of `mix_signals` for float32 CUDA tensors (B, N), (B, N_mu) and B target SMRs."""
    from .. import silence as _sil
    return _sil.mix_signals(Xin_sp, Xin_mu, target_dB)


def _read_audio(fName, sr=16000):
    """Minimal loader for the 'next' row: .npy (float array already at 16 kHz) or PCM/float .wav."""
    if fName.endswith(".npy"):
        return np.load(fName).astype(np.float32), sr
    from scipy.io import wavfile
    fs, x = wavfile.read(fName)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    if fs != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(fs), int(sr))
        x = resample_poly(x, sr // g, fs // g).astype(np.float32)
    return x, sr


def preprocess_signal_batch(Xin, fs, Tw, Ts):
    """preprocessing.py:332-349 for a float32 CUDA tensor (B, N) of equal-length clips, device resident.
    Returns (Xin_silrem (B, N), n_keep (B,)): like the reference the output keeps the input length, retained
    samples first, then the normalised image of its tail of ones."""
    from .. import silence as _sil
    return _sil.preprocess_signal(Xin, fs, Tw, Ts)


def load_and_preprocess_signal(fName, Tw, Ts):
    """preprocessing.py:330-350 -> (Xin_silrem float32, fs)."""
    Xin, fs = _read_audio(fName)
    dx = torch.from_numpy(np.ascontiguousarray(Xin, dtype=np.float32)).cuda()
    out, _ = preprocess_signal_batch(dx[None], fs, Tw, Ts)
    Xin_silrem = out[0].cpu().numpy()
    if len(Xin_silrem) / fs < 0.1:  # :343-346; duplication keeps mean and max, so the order with :348-349 is free
        while len(Xin_silrem) / fs < 0.1:
            Xin_silrem = np.append(Xin_silrem, Xin_silrem)
    return Xin_silrem, fs


# ---- featuregram ------------------------------------------------------------------------------------------
def featuregram_batch(PARAMS, Xin, n_fft, n_mels, featName, W=None, shift=None, taps=False, fs=16000):
    """Device fast path: Xin float32 CUDA tensor (B, n_samples) of equal-length clips ->
    dict(fv=(B, 2*rows, T)[, patches=(B*nP, W, 2*rows) time-major, standardised])."""
    cfg = _fe.FrontendConfig.from_params(PARAMS, n_fft, n_mels, featName, fs)
    return _frontend_for(cfg).run(Xin, W=W, shift=shift, taps=taps)


def featuregram_from_signal(PARAMS, Xin, n_fft, n_mels, featName, fs=16000):
    """One clip: the arithmetic of get_featuregram from `Xin` on (preprocessing.py:404-444)."""
    x = torch.from_numpy(np.ascontiguousarray(Xin, dtype=np.float32)).cuda()[None]
    return featuregram_batch(PARAMS, x, n_fft, n_mels, featName, fs=fs)["fv"][0].cpu().numpy()


def get_featuregram(PARAMS, classname, feature_opDir, fName_path_sp, fName_path_mu, target_dB, n_fft, n_mels,
                    featName, save_feat=True):
    """preprocessing.py:355-457: same naming, same .npy cache layout <feature_opDir>/<class>/<name>.npy."""
    if (fName_path_sp != '') and (fName_path_mu != ''):
        fName = (fName_path_sp.split('/')[-1].split('.')[0] + '_' + fName_path_mu.split('/')[-1].split('.')[0]
                 + '_' + str(target_dB) + 'dB')
    elif fName_path_sp != '':
        fName = fName_path_sp.split('/')[-1].split('.')[0]
    elif fName_path_mu != '':
        fName = fName_path_mu.split('/')[-1].split('.')[0]
    else:
        raise ValueError("get_featuregram: both file paths are empty")
    cache = feature_opDir + '/' + classname + '/' + fName + '.npy'
    if os.path.exists(cache):
        return np.load(cache, allow_pickle=False)
    if featName not in _fe.FEATS:
        raise ValueError("featName %r: only the HPSS features %s are on the built path" % (featName, sorted(_fe.FEATS)))
    if classname == 'speech_music':
        Xin_sp, fs = load_and_preprocess_signal(fName_path_sp, PARAMS['Tw'], PARAMS['Ts'])
        Xin_mu, fs = load_and_preprocess_signal(fName_path_mu, PARAMS['Tw'], PARAMS['Ts'])
        Xin = mix_signals(Xin_sp, Xin_mu, target_dB)
    elif classname in ('speech', 'muspeak'):
        Xin, fs = load_and_preprocess_signal(fName_path_sp, PARAMS['Tw'], PARAMS['Ts'])
    elif classname == 'music':
        Xin, fs = load_and_preprocess_signal(fName_path_mu, PARAMS['Tw'], PARAMS['Ts'])
    else:
        raise ValueError("unknown classname %r" % classname)
    fv = featuregram_from_signal(PARAMS, Xin, n_fft, n_mels, featName, fs)
    if save_feat:
        os.makedirs(feature_opDir + '/' + classname + '/', exist_ok=True)
        np.save(cache, fv)
    return fv


# ---- patches ------------------------------------------------------------------------------------------------
def get_feature_patches(PARAMS, FV, patch_size, patch_shift, featName):
    """preprocessing.py:137-292.  FV (nFeatures, nFrames) -> float64 (nP, F, W) (Lemaire models) or
    (nP, F, W, 1).  Tiling-if-short, the H/P split, StandardScaler and the patch gather run on the GPU."""
    FV = np.asarray(FV)
    if FV.ndim != 2:
        raise ValueError("FV should be of the shape (nFeatures, nFrames)")
    fe = _frontend_for(_fe.FrontendConfig())
    d = torch.from_numpy(np.ascontiguousarray(FV, dtype=np.float32)).cuda()
    F, T = d.shape
    scale = not PARAMS['frame_level_scaling']

    def run(rows):  # rows: (r, T) device tensor -> (nP, r, W) device tensor
        x = fe.standardize_rows(rows) if scale else rows
        return fe.extract_patches(x[None], patch_size, patch_shift, time_major=False)

    if featName in ('Spec', 'LogSpec', 'MelSpec', 'LogMelSpec'):
        patches = run(d)
    else:
        known = ('MelHarm', 'MelPerc', 'LogMelHarm', 'LogMelPerc', 'Harm', 'Perc', 'LogHarm', 'LogPerc')
        if not featName.startswith(known):
            raise ValueError("unknown featName %r" % featName)
        half = int(F / 2)
        base = featName.replace('LogMel', '').replace('Mel', '').replace('Log', '')  # HarmSpec/PercSpec/HarmPercSpec
        parts = []
        if base in ('HarmSpec', 'HarmPercSpec'):
            parts.append(run(d[:half]))
        if base in ('PercSpec', 'HarmPercSpec'):
            parts.append(run(d[half:]))
        if not parts:
            raise ValueError("unknown featName %r" % featName)
        patches = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
    patches = patches.cpu().numpy().astype(np.float64)
    if 'Lemaire_et_al' not in PARAMS['Model']:
        patches = np.expand_dims(patches, axis=3)
    return patches
