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
    """preprocessing.py:297-325 for one pair of signals: a batch of one through `smh_mix_signals_f32` (music looped / cut to
    the speech length, scaled to the target speech-to-music ratio, factors normalised to sum 1, then `normalize_signal`)."""
    sp = torch.from_numpy(np.ascontiguousarray(Xin_sp, dtype=np.float32)).cuda()[None]
    mu = torch.from_numpy(np.ascontiguousarray(Xin_mu, dtype=np.float32)).cuda()[None]
    return mix_signals_batch(sp, mu, [float(target_dB)])[0].cpu().numpy()


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


def audio_num_samples(fName, sr=16000):
    """Samples `_read_audio` will return for this file (after resampling to sr), from the header alone where possible."""
    if fName.endswith(".npy"):
        return int(np.load(fName, mmap_mode="r").shape[0])
    import wave
    try:
        with wave.open(fName, "rb") as w:
            n, fs = w.getnframes(), w.getframerate()
    except (wave.Error, EOFError):  # float / extensible WAV: let scipy parse it
        x, _ = _read_audio(fName, sr)
        return int(len(x))
    if fs == sr:
        return int(n)
    from math import gcd
    g = gcd(int(fs), int(sr))
    up, down = sr // g, fs // g
    return int(-(-n * up // down))  # scipy.signal.resample_poly: ceil(n * up / down)


def feature_cache_path(feature_opDir, classname, fName_path_sp, fName_path_mu, target_dB):
    """<feature_opDir>/<classname>/<name>.npy as get_featuregram names it (preprocessing.py:357-363)."""
    if (fName_path_sp != '') and (fName_path_mu != ''):
        fName = (fName_path_sp.split('/')[-1].split('.')[0] + '_' + fName_path_mu.split('/')[-1].split('.')[0]
                 + '_' + str(target_dB) + 'dB')
    elif fName_path_sp != '':
        fName = fName_path_sp.split('/')[-1].split('.')[0]
    elif fName_path_mu != '':
        fName = fName_path_mu.split('/')[-1].split('.')[0]
    else:
        raise ValueError("get_featuregram: both file paths are empty")
    return feature_opDir + '/' + classname + '/' + fName + '.npy'


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
    cache = feature_cache_path(feature_opDir, classname, fName_path_sp, fName_path_mu, target_dB)
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
