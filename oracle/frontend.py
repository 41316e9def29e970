"""numpy restatement of the reference's feature front end (TEST INFRASTRUCTURE, see oracle/__init__).

Every function cites the reference lines (``/root/reference`` paths) whose arithmetic it follows.
Where the arithmetic lives in librosa (not vendored, not installable offline, version unpinned --
call style implies 0.7 <= librosa < 0.10, most likely 0.8.x) the published librosa-0.8 algorithm is
restated and marked [librosa]; those parts are "parity unpinned".

Conventions: spectrograms are (K, T) = (bins, frames), float32, exactly as librosa returns them.
"""
from __future__ import annotations

import numpy as np

F32_TINY = np.finfo(np.float32).tiny


# --------------------------------------------------------------------------------------------------
# signal conditioning  (lib/preprocessing.py:114-132, 297-325)
# --------------------------------------------------------------------------------------------------
def normalize_signal(x):
    """lib/preprocessing.py:130-131 -- mean removal then peak normalisation (dtype preserved)."""
    x = x - np.mean(x)
    x = x / np.max(np.abs(x))
    return x


def mix_signals(x_sp, x_mu, target_db):
    """lib/preprocessing.py:297-325 -- loop music to speech length, scale to target SMR, renormalise."""
    mu = x_mu.copy()
    while len(mu) < len(x_sp):
        mu = np.append(mu, x_mu)
    n = min(len(x_sp), len(mu))
    sp, mu = x_sp[:n], mu[:n]
    e_sp = np.sum(np.power(sp, 2)) / len(sp)
    e_mu = np.sum(np.power(mu, 2)) / len(mu)
    req = e_sp / np.power(10, (target_db / 10))
    f_mu = np.sqrt(req / e_mu)
    f_sp = 1
    s = f_mu + f_sp
    f_mu /= s
    f_sp /= s
    # numpy 1.x value-based casting kept float32 audio float32 here; be explicit about it
    mix = (np.float32(f_sp) * sp + np.float32(f_mu) * mu).astype(sp.dtype)
    return normalize_signal(mix)


# --------------------------------------------------------------------------------------------------
# a1  STFT magnitude   (call sites lib/preprocessing.py:407,417,429,439)  [librosa.core.stft]
# --------------------------------------------------------------------------------------------------
def hann_window(win_length: int, n_fft: int) -> np.ndarray:
    """[librosa] get_window('hann', win_length, fftbins=True) -> periodic Hann, float64,
    zero-padded symmetrically to n_fft (util.pad_center) when win_length < n_fft.
    Evaluated the way scipy.signal.get_window does it (general_cosine on linspace(-pi, pi, M + 1), last point dropped), so the
    table is BIT-identical to the routine librosa delegates to (tests/test_oracle_pins.py); the textbook form
    0.5 - 0.5 cos(2 pi n / M) differs from it by up to 8e-16."""
    fac = np.linspace(-np.pi, np.pi, win_length + 1)
    w = (0.5 + 0.5 * np.cos(fac))[:-1]
    if win_length < n_fft:
        lpad = (n_fft - win_length) // 2
        w = np.pad(w, (lpad, n_fft - win_length - lpad))
    return w


def num_frames(n_samples: int, n_fft: int, hop: int) -> int:
    """[librosa] util.frame with center=False: 1 + (N - n_fft)//hop  (bit-exact integer contract)."""
    if n_samples < n_fft:
        return 0
    return 1 + (n_samples - n_fft) // hop


def stft_mag(y: np.ndarray, n_fft: int = 400, win_length: int = 400, hop: int = 160) -> np.ndarray:
    """np.abs(librosa.core.stft(y, n_fft, win_length=, hop_length=, center=False)).

    [librosa 0.8] frame t = y[t*hop : t*hop+n_fft]; float64 window * float32 frame -> float64 rfft ->
    stored as complex64; np.abs of complex64 -> float32.  Returns (1+n_fft//2, T) float32.
    """
    y = np.asarray(y)
    T = num_frames(len(y), n_fft, hop)
    w = hann_window(win_length, n_fft)
    idx = np.arange(n_fft)[:, None] + hop * np.arange(T)[None, :]
    frames = y[idx]  # (n_fft, T)
    spec = np.fft.rfft(w[:, None] * frames, axis=0).astype(np.complex64)
    return np.abs(spec)  # float32


# --------------------------------------------------------------------------------------------------
# a2  median filters   [librosa.decompose.hpss -> scipy.ndimage.median_filter(mode='reflect')]
# --------------------------------------------------------------------------------------------------
def _reflect_index(i: np.ndarray, n: int) -> np.ndarray:
    """scipy.ndimage 'reflect' == numpy.pad 'symmetric': (d c b a | a b c d | d c b a), period 2n."""
    j = np.mod(i, 2 * n)
    return np.where(j >= n, 2 * n - 1 - j, j)


def median_filter_1d(S: np.ndarray, size: int, axis: int) -> np.ndarray:
    """Sliding median of odd ``size`` along ``axis`` with 'reflect' boundary.  Pure selection, so
    bit-exact against scipy is required (tests/test_oracle_pins.py).  Independent of scipy: gathers
    the window by index and takes the middle of a sort."""
    assert size % 2 == 1 and size >= 1
    h = size // 2
    n = S.shape[axis]
    pos = np.arange(n)[:, None] + np.arange(-h, h + 1)[None, :]  # (n, size)
    pos = _reflect_index(pos, n)
    Sm = np.moveaxis(S, axis, -1)
    win = Sm[..., pos]  # (..., n, size)
    med = np.sort(win, axis=-1)[..., h]
    return np.moveaxis(med, -1, axis)


def median_time(S, l_harm):
    """harm = median_filter(S, size=(1, l_harm), mode='reflect')  -- along frames (axis 1)."""
    return median_filter_1d(S, l_harm, axis=-1)


def median_freq(S, l_perc):
    """perc = median_filter(S, size=(l_perc, 1), mode='reflect')  -- along bins (axis 0)."""
    return median_filter_1d(S, l_perc, axis=-2)


# --------------------------------------------------------------------------------------------------
# a3  softmask   [librosa.util.softmask(X, X_ref, power=2, split_zeros=True)]
# --------------------------------------------------------------------------------------------------
def softmask(X: np.ndarray, X_ref: np.ndarray) -> np.ndarray:
    """power=2, split_zeros=True (hpss defaults margin=1 -> split_zeros).  All float32, op order kept:
    Z=max(X,Xref); bad=Z<tiny -> Z=1; m=(X/Z)^2; r=(Xref/Z)^2; m=m/(m+r); m[bad]=0.5."""
    X = X.astype(np.float32, copy=False)
    X_ref = X_ref.astype(np.float32, copy=False)
    Z = np.maximum(X, X_ref)
    bad = Z < F32_TINY
    Z = np.where(bad, np.float32(1), Z)
    a = X / Z
    b = X_ref / Z
    m = a * a
    r = b * b
    with np.errstate(invalid="ignore", divide="ignore"):
        out = m / (m + r)
    out = np.where(bad, np.float32(0.5), out)
    return out.astype(np.float32)


def hpss(S: np.ndarray, l_harm: int, l_perc: int):
    """[librosa.decompose.hpss(S, kernel_size=(l_harm, l_perc))] power=2, margin=1, mask=False.
    Returns (H, P, harm, perc), all (K, T) float32."""
    S = S.astype(np.float32, copy=False)
    harm = median_time(S, l_harm)
    perc = median_freq(S, l_perc)
    mask_h = softmask(harm, perc)
    mask_p = softmask(perc, harm)
    return S * mask_h, S * mask_p, harm, perc


# --------------------------------------------------------------------------------------------------
# a4  mel filterbank   [librosa.filters.mel, librosa.feature.melspectrogram(S=...)]
# --------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_basis(sr: float = 22050, n_fft: int = 400, n_mels: int = 120) -> np.ndarray:
    """[librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney')] float32.

    NB the reference calls melspectrogram(S=..., n_mels=...) WITHOUT sr (lib/preprocessing.py:409-410,
    419,421), so librosa's default sr=22050 builds the basis although the audio is 16 kHz."""
    fmax = float(sr) / 2
    K = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, K, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    W = np.zeros((n_mels, K), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        W[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    W *= enorm[:, np.newaxis]
    return W


def mel_project(S: np.ndarray, n_mels: int) -> np.ndarray:
    """melspectrogram(S=S, n_mels=n_mels): n_fft inferred 2*(K-1); no power applied; basis @ S."""
    K = S.shape[0]
    B = mel_basis(22050, 2 * (K - 1), n_mels)
    # float64 accumulate then round: a BLAS-order-independent statement of np.dot(float32, float32)
    return (B.astype(np.float64) @ S.astype(np.float64)).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# a5  power_to_db   [librosa.core.power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0)]
# --------------------------------------------------------------------------------------------------
def power_to_db(S: np.ndarray) -> np.ndarray:
    """10*log10(max(amin,S)) - 10*log10(max(amin,1)); then max(., max_over_whole_array - 80). f32."""
    S = np.asarray(S, dtype=np.float32)
    log_spec = np.float32(10.0) * np.log10(np.maximum(np.float32(1e-10), S))
    log_spec = log_spec - np.float32(0.0)
    return np.maximum(log_spec, log_spec.max() - np.float32(80.0)).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# get_featuregram HPSS branches   (lib/preprocessing.py:404-444)
# --------------------------------------------------------------------------------------------------
HPSS_FEATS = ("MelHarmPercSpec", "LogMelHarmPercSpec", "HarmPercSpec", "LogHarmPercSpec")


def featuregram(y: np.ndarray, featName: str = "LogMelHarmPercSpec", n_fft: int = 400, n_mels: int = 120,
                l_harm: int = 21, l_perc: int = 11, Tw: int = 25, Ts: int = 10, fs: int = 16000,
                return_parts: bool = False):
    """The four '*HarmPerc*' branches of get_featuregram starting at Xin (file I/O, silence removal
    and caching stripped): returns fv = [fv_H ; fv_P] stacked on axis 0, float32."""
    frameSize = int(Tw * fs / 1000)
    frameShift = int(Ts * fs / 1000)
    S = stft_mag(y, n_fft=n_fft, win_length=frameSize, hop=frameShift)
    return featuregram_from_S(S, featName, n_mels=n_mels, l_harm=l_harm, l_perc=l_perc, return_parts=return_parts)


def featuregram_from_S(S: np.ndarray, featName: str = "LogMelHarmPercSpec", n_mels: int = 120, l_harm: int = 21,
                       l_perc: int = 11, return_parts: bool = False):
    """Everything of the '*HarmPerc*' branches behind `np.abs(librosa.core.stft(...))`
    (lib/preprocessing.py:408-412, 418-424, 430-434, 440-444), starting from a GIVEN magnitude spectrogram.
    Parity tests feed the device's own S here: the medians are then selections of identical values, so the
    dB outputs can be held to abs 1e-3 on every bin (no selection flips caused by the STFT's last ulp)."""
    S = np.asarray(S, dtype=np.float32)
    H, P, harm, perc = hpss(S, l_harm, l_perc)
    if featName.startswith("MelHarm"):  # :404-412
        fv_H, fv_P = mel_project(H, n_mels), mel_project(P, n_mels)
    elif featName.startswith("LogMelHarm"):  # :414-424
        fv_H = power_to_db(mel_project(H, n_mels) ** 2)
        fv_P = power_to_db(mel_project(P, n_mels) ** 2)
    elif featName.startswith("Harm"):  # :426-434
        fv_H, fv_P = H, P
    elif featName.startswith("LogHarm"):  # :436-444
        fv_H, fv_P = power_to_db(H ** 2), power_to_db(P ** 2)
    else:
        raise ValueError(featName)
    fv = np.append(fv_H, fv_P, axis=0).astype(np.float32)
    if return_parts:
        return fv, dict(S=S, harm=harm, perc=perc, H=H, P=P)
    return fv


# --------------------------------------------------------------------------------------------------
# a7 / a8  get_feature_patches + tools.extract_patches
# --------------------------------------------------------------------------------------------------
def tile_if_short(FV: np.ndarray, patch_size: int) -> np.ndarray:
    """lib/preprocessing.py:139-142 -- `if T < W: while T <= W: FV = [FV, FV1]` (note < then <=)."""
    if FV.shape[1] < patch_size:
        FV1 = FV.copy()
        while FV.shape[1] <= patch_size:
            FV = np.append(FV, FV1, axis=1)
    return FV


def standardize_rows(FV: np.ndarray) -> np.ndarray:
    """StandardScaler(copy=False).fit_transform(FV.T).T (lib/preprocessing.py:211-214): per-row mean
    and population std over frames; std==0 -> 1.  sklearn accumulates in float64 and applies
    `X -= mean; X /= scale` onto the float32 array (two roundings)."""
    X = FV.astype(np.float32, copy=True)
    x64 = X.astype(np.float64)
    mean = x64.mean(axis=1)
    var = ((x64 - mean[:, None]) ** 2).mean(axis=1)
    scale = np.sqrt(var)
    # sklearn _handle_zeros_in_scale / _is_constant_feature: (near-)constant rows are not scaled
    n = X.shape[1]
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2
    scale = np.where(constant | (scale == 0.0), 1.0, scale)
    X = (x64 - mean[:, None]).astype(np.float32)
    X = (X.astype(np.float64) / scale[:, None]).astype(np.float32)
    return X


def patch_starts(n_frames: int, patch_size: int, patch_shift: int) -> list:
    """lib/cython_impl/tools.pyx:24-34 -- centre i in range(half, T-half, shift); start=i-half;
    end=min(start+W,T); if short start=end-W.  Integer contract, must be bit-exact."""
    half = int(patch_size / 2)
    starts = []
    for i in range(half, n_frames - half, patch_shift):
        s = i - half
        e = min(s + patch_size, n_frames)
        if e - s < patch_size:
            s = e - patch_size
        starts.append(s)
    return starts


def extract_patches(FV: np.ndarray, patch_size: int, patch_shift: int) -> np.ndarray:
    """lib/cython_impl/tools.pyx:21-38 -> float64 (nP, F, W)."""
    starts = patch_starts(FV.shape[1], patch_size, patch_shift)
    out = np.zeros((len(starts), FV.shape[0], patch_size))
    for k, s in enumerate(starts):
        out[k] = FV[:, s : s + patch_size]
    return out


def feature_patches(FV: np.ndarray, patch_size: int, patch_shift: int, featName: str = "LogMelHarmPercSpec",
                    lemaire: bool = True) -> np.ndarray:
    """get_feature_patches, '*HarmPerc*' branches with frame_level_scaling False
    (lib/preprocessing.py:137-142 + 180-290): tile, split halves, standardise each, patch each,
    concatenate on the feature axis; non-Lemaire models get a trailing channel axis."""
    FV = tile_if_short(FV, patch_size)
    half = int(FV.shape[0] / 2)
    parts = []
    want_h = "Harm" in featName
    want_p = "Perc" in featName
    if want_h:
        parts.append(extract_patches(standardize_rows(FV[:half]), patch_size, patch_shift))
    if want_p:
        parts.append(extract_patches(standardize_rows(FV[half:]), patch_size, patch_shift))
    patches = np.concatenate(parts, axis=1) if len(parts) > 1 else parts[0].copy()
    if not lemaire:
        patches = np.expand_dims(patches, axis=3)
    return patches


def tcn_input(patches: np.ndarray) -> np.ndarray:
    """Proposed_Work_Results.py:235-236,483-484 -- (N,F,W) -> (N,W,F); Keras then casts to float32."""
    return np.transpose(patches, (0, 2, 1)).astype(np.float32)
