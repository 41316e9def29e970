"""Counterparts of the reference's two data generators, the callers of the hot path:

  generator(PARAMS, folder, file_list, batchSize)                       Proposed_Work_Results.py:49-270
  test_file_wise_generator(PARAMS, file_name_sp, file_name_mu, target_dB)  Proposed_Work_Results.py:459-496

Same arguments, same PARAMS keys, same batch composition and label rules, same use of numpy's global random state
(`np.random.shuffle` / `np.random.choice`: seed it the way the reference's driver would).  What differs is WHERE the work
happens: the reference computes one file at a time on the CPU and grows its buffers with `np.append`; here the files a
batch needs are decided first -- the number of patches a file yields is an integer contract of its length
(`smh_num_frames` / `smh_tiled_frames` / `smh_num_patches`), and silence removal keeps the length -- and then go through
the HIP front end together as ONE ragged device batch (`Frontend.run_ragged`); patches stay on the device as float32
tensors in the network's (N, W, 2F) layout.  Per-file results do not depend on what else is in the batch, so the yielded
batches are the ones the sequential loop would yield.

Data parallel (SURVEY 8e "class-balanced batch composition must be preserved globally, not per rank"): `generator(...,
rank=, world=)` -- defaulting to the initialised torch.distributed process group -- builds the SAME globally class-balanced
batch of 3 * batchSize rows on every rank (same file lists, same numpy random state: the caller seeds numpy identically on all
ranks, exactly as a single process would be seeded; nothing is seeded here) and hands rank r the rows
`sharding.shard_indices(3 * batchSize, r, world)` of it with their labels: the union over the ranks IS the single-process
batch, row for row.  Ranks whose numpy state has drifted apart would silently train on different "global" batches, so the
state is compared across the ranks on the first batch and every `check_every` batches (one tiny all-reduce) and a mismatch
raises.

`featuregram_fn` / `patches_fn` are injection points for tests (and for a CPU oracle): per-file callables with the
signatures of `preproc.get_featuregram` / `preproc.get_feature_patches`.  When they are given, files are processed one
by one through them, exactly like the reference's loop.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from . import batching

HARDCODED_TEST_SHIFT = 68  # Proposed_Work_Results.py:474 passes 68 as patch_shift whatever PARAMS['W_shift'] says


def to_categorical(labels, num_classes):
    """tensorflow.keras.utils.to_categorical for a 1-D integer label list (float32 one-hot)."""
    labels = np.asarray(labels, dtype=np.int64).reshape(-1)
    out = np.zeros((labels.size, int(num_classes)), dtype=np.float32)
    out[np.arange(labels.size), labels] = 1.0
    return out


class _ClassBuffer:
    """One class's queue of patches, file by file, in the order the reference's loop would append them.  A file enters with the
    NUMBER of patches it will yield (an integer contract of its length) and, once somebody needs them, its patches.  `take(n, lo, hi)`
    pops the first n rows of the queue -- the class's share of the global batch -- and returns rows lo .. hi-1 of them: the whole
    share in a single process, this rank's contiguous range under data parallel, so that a rank computes only the files its rows
    come from (`pending(lo, hi)` names them)."""

    def __init__(self):
        self.files = []  # [spec, meta, n_rows_left, first_row_left, patches or None]

    @property
    def balance(self):
        return sum(f[2] for f in self.files)

    def add(self, spec, meta, n, patches=None):
        if patches is not None:
            n = int(patches.shape[0])
        if n > 0:
            self.files.append([spec, meta, int(n), 0, patches])

    def _overlapping(self, lo, hi):
        pos = 0
        for f in self.files:
            a, b = pos, pos + f[2]
            if a < hi and b > lo:
                yield f, max(lo - a, 0), min(hi, b) - a  # the file's rows [u, v) of what is left of it
            pos = b
            if pos >= hi:
                break

    def pending(self, lo, hi):
        """Files among the queue's rows lo .. hi-1 whose patches have not been computed yet."""
        return [f for f, _, _ in self._overlapping(lo, hi) if f[4] is None]

    def set_patches(self, f, patches, strict):
        got = int(patches.shape[0])
        want = f[2] + f[3]
        if got != want:
            if strict:
                raise RuntimeError("file %r yields %d patches where its length promised %d: under data parallel the ranks decide a "
                                   "batch's files from the promised counts and would drift apart" % (f[0], got, want))
            f[2] = max(got - f[3], 0)
        f[4] = patches

    def take(self, n, lo=0, hi=None):
        hi = n if hi is None else hi
        parts = [f[4][f[3] + u:f[3] + v] for f, u, v in self._overlapping(lo, hi)]
        meta, left = [], n
        while left > 0:  # pop the first n rows of the queue
            f = self.files[0]
            k = min(left, f[2])
            meta.extend([f[1]] * k)
            f[2] -= k
            f[3] += k
            left -= k
            if f[2] == 0:
                self.files.pop(0)
        if isinstance(parts[0], np.ndarray):
            data = np.concatenate(parts, axis=0)
        else:
            import torch
            data = torch.cat(parts, dim=0)
        return data, meta


# ---- device-resident featuregram cache ------------------------------------------------------------------------------------
# The reference caches every file's featuregram as <feature_opDir>/<class>/<name>.npy (lib/preprocessing.py:357-363, 446-449) and
# re-reads it from disk in every later epoch.  An MI355X has 288 GB of HBM: MUSAN's 163.5 h at 100 frames/s x 240 rows x 4 bytes are
# 56 GB -- the whole corpus fits beside the model.  A cached featuregram is therefore uploaded ONCE and kept on the device (LRU under
# a byte budget); later batches start from the device copy: no np.load, no host-to-device copy -- and no host synchronisation, which
# a pageable upload is, so the host can run ahead of the device and fit()'s side stream has something to overlap.  The key carries the
# file's size and modification time: a rewritten cache file is read again.  SMH_FV_CACHE_GB sets the budget (0 disables; default a
# quarter of the device's memory), and an entry is only admitted while at least as much memory again stays FREE on the device
# (torch.cuda.mem_get_info at insert time: several ranks or another tenant on the same GPU shrink the cache instead of running
# out of memory).
_FV_CACHE = OrderedDict()
_FV_CACHE_BYTES = [0]


def _fv_cache_budget():
    ev = os.environ.get("SMH_FV_CACHE_GB")
    if ev is not None:
        return int(float(ev) * 2 ** 30)
    import torch
    return torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory // 4


def fv_cache_clear():
    """Drop every device-resident featuregram (tests; or to hand the memory back)."""
    _FV_CACHE.clear()
    _FV_CACHE_BYTES[0] = 0


def _cached_featuregram(path, fresh=None):
    """The .npy featuregram at `path` as a float32 device tensor, from the device cache when it is there.
    fresh: the device tensor that has just been written to `path` (the file need not be read back to enter the cache)."""
    import torch
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size, torch.cuda.current_device())
    t = _FV_CACHE.get(key)
    if t is not None:
        _FV_CACHE.move_to_end(key)
        return t
    if fresh is not None:
        t = fresh.detach().to(dtype=torch.float32).clone()  # (a copy: the ragged pass hands out views of one buffer)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.load(path, allow_pickle=False), dtype=np.float32)).cuda()
    budget, nbytes = _fv_cache_budget(), t.numel() * 4
    if os.environ.get("SMH_FV_CACHE_GB") is None:
        try:  # default budget: never more than what keeps half of the currently free memory free
            free, _ = torch.cuda.mem_get_info()
            budget = min(budget, _FV_CACHE_BYTES[0] + free // 2)
        except RuntimeError:
            pass
    if nbytes <= budget:
        cur = torch.cuda.current_stream()
        while _FV_CACHE and _FV_CACHE_BYTES[0] + nbytes > budget:
            _, old = _FV_CACHE.popitem(last=False)
            _FV_CACHE_BYTES[0] -= old.numel() * 4
            # the evicted tensor may still be read by work enqueued on this stream (fit's side stream builds batches from the
            # cache): the allocator must not hand its memory out before that work is done -- record_stream instead of a
            # device-wide synchronize in the middle of building a batch
            old.record_stream(cur)
        _FV_CACHE[key] = t
        _FV_CACHE_BYTES[0] += nbytes
    return t


def _device_patches_for(PARAMS, specs, featName, n_fft, n_mels, W, shift):
    """specs: list of (classname, sp_path, mu_path, target_dB).  Loads / conditions / mixes the signals (the 'next' row in
    front of the path: lib.preprocessing.load_and_preprocess_signal, mix_signals), then ONE ragged pass of the front end.
    Returns a list of (nP_i, W, 2F) float32 device tensors."""
    import torch
    from . import frontend as _fe
    from .lib import preprocessing as pp
    if 'Lemaire_et_al' not in PARAMS['Model']:
        raise ValueError("the device generator yields the TCN layout (N, W, F): Model must be a Lemaire_et_al variant")
    clips = []
    for classname, sp, mu, db in specs:
        cache = pp.feature_cache_path(PARAMS['feature_opDir'], classname, sp, mu, db)
        if os.path.exists(cache):
            clips.append(("fv", cache))
            continue
        if classname == 'speech_music':
            x_sp, _ = pp.load_and_preprocess_signal(sp, PARAMS['Tw'], PARAMS['Ts'])
            x_mu, _ = pp.load_and_preprocess_signal(mu, PARAMS['Tw'], PARAMS['Ts'])
            x = pp.mix_signals(x_sp, x_mu, db)
        elif classname == 'speech':
            x, _ = pp.load_and_preprocess_signal(sp, PARAMS['Tw'], PARAMS['Ts'])
        else:
            x, _ = pp.load_and_preprocess_signal(mu, PARAMS['Tw'], PARAMS['Ts'])
        clips.append(("audio", np.ascontiguousarray(x, dtype=np.float32), cache))
    cfg = _fe.FrontendConfig.from_params(PARAMS, n_fft, n_mels, featName)
    fe = pp._frontend_for(cfg)
    out = [None] * len(clips)
    audio_idx = [i for i, c in enumerate(clips) if c[0] == "audio"]
    if audio_idx:
        res = fe.run_ragged([clips[i][1] for i in audio_idx], W=W, shift=shift)
        for k, i in enumerate(audio_idx):
            out[i] = res["patches"][k]
            if PARAMS.get('save_features', True):  # get_featuregram(save_feat=True): the reference's .npy cache
                os.makedirs(os.path.dirname(clips[i][2]), exist_ok=True)
                tmp = "%s.%d.tmp.npy" % (clips[i][2], os.getpid())  # (two ranks may compute the same file: whole files only)
                np.save(tmp, res["fv"][k].cpu().numpy())
                os.replace(tmp, clips[i][2])
                _cached_featuregram(clips[i][2], fresh=res["fv"][k])  # ... and stays on the device for the epochs to come
    for i, c in enumerate(clips):
        if c[0] == "fv":  # cached featuregram (device copy after its first use): standardise + patches only
            out[i] = fe.patches_from_featuregram(_cached_featuregram(c[1]), W, shift)
    return out


def _n_patches_of_file(PARAMS, path, n_fft, W, shift, lengths_cache):
    """Patches a file will yield, from its length alone (bit-exact integer contracts; silence removal keeps the length)."""
    from . import _lib
    from .lib import preprocessing as pp
    n = lengths_cache.get(path)
    if n is None:
        n = lengths_cache[path] = pp.audio_num_samples(path)
    lib = _lib.load()
    fs = 16000
    hop = int(PARAMS['Ts'] * fs / 1000)
    if n / fs < 0.1:  # load_and_preprocess_signal doubles clips shorter than 0.1 s (preprocessing.py:343-346)
        while n / fs < 0.1:
            n *= 2
    T = lib.smh_num_frames(int(n), int(n_fft), hop)
    if T < 1:
        return 0
    return lib.smh_num_patches(lib.smh_tiled_frames(T, int(W)), int(W), int(shift))


def _n_patches_of_spec(PARAMS, spec, n_fft, n_mels, W, shift, lengths_cache):
    """Patches the file(s) of `spec` will yield, before anything is computed: from the cached featuregram's header when the
    reference's .npy cache holds it (its frame count is what `patches_from_featuregram` will see), else from the audio's length
    (a mixture has the length of its speech file: the music is looped / cut to it, preprocessing.py:303-310)."""
    from . import _lib
    from .lib import preprocessing as pp
    classname, sp, mu, db = spec
    cache = pp.feature_cache_path(PARAMS['feature_opDir'], classname, sp, mu, db)
    key = ("fv", cache)
    if key in lengths_cache:
        return lengths_cache[key]
    if os.path.exists(cache):
        try:
            T = int(np.load(cache, mmap_mode="r", allow_pickle=False).shape[1])
            lib = _lib.load()
            n = lib.smh_num_patches(lib.smh_tiled_frames(T, int(W)), int(W), int(shift)) if T >= 1 else 0
            lengths_cache[key] = n
            return n
        except (OSError, ValueError, IndexError):
            pass
    return _n_patches_of_file(PARAMS, mu if classname == 'music' else sp, n_fft, W, shift, lengths_cache)


def _rank_world(rank, world):
    """(rank, world, dist): explicit arguments, else the initialised process group, else a single process."""
    dist = None
    try:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            dist = td
    except ImportError:  # pragma: no cover
        pass
    if world is None:
        world = dist.get_world_size() if dist is not None else 1
    if rank is None:
        rank = dist.get_rank() if dist is not None else 0
    if not (0 <= int(rank) < int(world)):
        raise ValueError("rank %r outside world %r" % (rank, world))
    if dist is not None and dist.get_world_size() != int(world):
        dist = None  # an explicit (rank, world) that is not the process group's: no cross-rank check possible
    return int(rank), int(world), dist


def _numpy_state_word():
    """A 52-bit digest of numpy's global random state (what decides every future shuffle / choice of the generator)."""
    import zlib
    st = np.random.get_state()
    return float((zlib.crc32(st[1].tobytes()) ^ (int(st[2]) * 2654435761)) & ((1 << 52) - 1))


def _check_same_state(dist, batch_count):
    import torch
    w = _numpy_state_word()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([w, -w], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    lo, hi = -float(t[1]), float(t[0])
    if lo != hi:
        raise RuntimeError("data-parallel generator: numpy's random state differs between the ranks at batch %d -- every rank must "
                           "seed numpy identically (the ranks build the same global batch and take their rows of it)" % batch_count)


def generator(PARAMS, folder, file_list, batchSize, featuregram_fn=None, patches_fn=None, rank=None, world=None, check_every=50,
              count_fn=None, batch_patches_fn=None):
    """Infinite class-balanced batch generator: yields (batchData, labels).
    rank / world (default: the torch.distributed process group, else one process): this rank's rows
    sharding.class_block_rows(n_classes, batchSize, rank, world) of every GLOBAL batch -- the same contiguous range of every class
    block; batchSize stays the global per-class batch size.  On the device path a rank runs the front end only for the files its
    rows come from (SURVEY 8e "Partitioning": shard per audio clip), decided from the files' patch counts before any kernel runs.
    count_fn(spec) / batch_patches_fn(specs): injection points of the device path for tests (patch count of a file from its length;
    the patches of a list of files as one batch) -- defaults: the integer contracts of libsmh and ONE ragged front-end pass.

    batchData: [batchSize music | batchSize speech | batchSize speech+music] patches, (3*batchSize, W, 2F) for the TCN
    models ((.., 2F, W, 1) for the Conv2D ones when the per-file callables are injected); labels: {'R', 'S', 'M', '3C'} for
    the MTL models (the S = M = 0 rule for mixtures included, Proposed_Work_Results.py:249-260), else the one-hot matrix."""
    batch_count = 0
    np.random.shuffle(file_list['speech'])
    np.random.shuffle(file_list['music'])
    file_list_sp_temp = file_list['speech'].copy()
    file_list_mu_temp = file_list['music'].copy()
    three = len(PARAMS['classes']) == 3
    if three:
        np.random.shuffle(file_list['speech+music'])
        file_list_spmu_temp = file_list['speech+music'].copy()
    for sub in ('speech', 'music') + (('speech_music',) if three else ()):
        os.makedirs(PARAMS['feature_opDir'] + '/' + sub + '/', exist_ok=True)
    n_fft = PARAMS['n_fft'][PARAMS['Model']]
    n_mels = PARAMS['n_mels'][PARAMS['Model']]
    featName = PARAMS['featName'][PARAMS['Model']]
    W, W_shift = PARAMS['W'], PARAMS['W_shift']
    if PARAMS.get('frame_level_scaling') or PARAMS.get('skewness_vector'):
        raise ValueError("frame_level_scaling / skewness_vector are off in every reference configuration of this path "
                         "(Proposed_Work_Results.py:802-804) and are not wired into the generator")
    from .sharding import class_block_rows, shard_range
    rank, world, dist = _rank_world(rank, world)
    if world > int(batchSize):
        raise ValueError("batchSize = %d rows per class cannot be shared out between %d ranks" % (batchSize, world))
    # this rank's rows of every class block: the same contiguous range [lo, hi) of the class's batchSize rows
    lo, hi = shard_range(int(batchSize), rank, world)
    per_file = featuregram_fn is not None or patches_fn is not None
    if per_file:
        from .lib import preprocessing as pp
        featuregram_fn = featuregram_fn or pp.get_featuregram
        patches_fn = patches_fn or pp.get_feature_patches
    lengths = {}
    buf = {'speech': _ClassBuffer(), 'music': _ClassBuffer(), 'speech_music': _ClassBuffer()}
    rng = np.random  # the reference draws from numpy's global state
    count_fn = count_fn or (lambda spec: _n_patches_of_spec(PARAMS, spec, n_fft, n_mels, W, W_shift, lengths))
    batch_patches_fn = batch_patches_fn or (lambda specs: _device_patches_for(PARAMS, specs, featName, n_fft, n_mels, W, W_shift))

    def one_file(spec):
        classname, sp, mu, db = spec
        fv = featuregram_fn(PARAMS, classname, PARAMS['feature_opDir'], sp, mu, db, n_fft, n_mels, featName)
        return patches_fn(PARAMS, fv, W, W_shift, featName)

    def fill(classname, next_spec):
        """Pop files until the class queue holds a batch (the reference's `while balance < batchSize` loops).  With the per-file
        callables a file's patches are computed as it is popped; on the device path only its patch COUNT is needed here -- an
        integer contract of its length -- and `compute` below runs the front end for the files this rank's rows come from."""
        b = buf[classname]
        while b.balance < batchSize:
            spec = next_spec()
            if spec is None:
                continue
            if per_file:
                b.add(spec, spec[3], 0, one_file(spec))
            else:
                b.add(spec, spec[3], count_fn(spec))

    def compute(classes_, next_of):
        """The front end for every file this rank's rows [lo, hi) of each class block come from and that has no patches yet: ONE
        ragged device batch for the whole global batch's share.  In a single process a file that turns out shorter than promised
        (a stale cache file) makes the class top itself up; under data parallel that is an error -- the ranks pop files by count."""
        while True:
            todo = [(c, f) for c in classes_ for f in buf[c].pending(lo, hi)]
            if not todo:
                return
            for (c, f), patches in zip(todo, batch_patches_fn([f[0] for _, f in todo])):
                buf[c].set_patches(f, patches, strict=world > 1)
            short = [c for c in classes_ if buf[c].balance < batchSize]
            if not short:
                return
            for c in short:
                fill(c, next_of[c])

    def next_speech():
        nonlocal file_list_sp_temp
        if not file_list_sp_temp:
            file_list_sp_temp = file_list['speech'].copy()
        path = folder + '/speech/' + file_list_sp_temp.pop()
        return ('speech', path, '', None) if os.path.exists(path) else None

    def next_music():
        nonlocal file_list_mu_temp
        if not file_list_mu_temp:
            file_list_mu_temp = file_list['music'].copy()
        path = folder + '/music/' + file_list_mu_temp.pop()
        return ('music', '', path, None) if os.path.exists(path) else None

    def next_mix():
        nonlocal file_list_spmu_temp
        if not file_list_spmu_temp:
            file_list_spmu_temp = file_list['speech+music'].copy()
        np.random.shuffle(file_list_spmu_temp)  # the reference reshuffles before every pop (:181)
        info = file_list_spmu_temp.pop()
        sp, mu = folder + '/speech/' + info['speech'], folder + '/music/' + info['music']
        return ('speech_music', sp, mu, info['SMR']) if (os.path.exists(sp) and os.path.exists(mu)) else None

    next_of = {'speech': next_speech, 'music': next_music, 'speech_music': next_mix if three else None}
    classes_ = ['music', 'speech'] + (['speech_music'] if three else [])
    mine = class_block_rows(len(classes_), int(batchSize), rank, world) if world > 1 else None
    while 1:
        fill('speech', next_speech)
        fill('music', next_music)
        if three:
            fill('speech_music', next_mix)
        if not per_file:
            compute(classes_, next_of)
        data_mu, _ = buf['music'].take(batchSize, lo, hi)
        data_sp, _ = buf['speech'].take(batchSize, lo, hi)
        parts = [data_mu, data_sp]
        smr = None
        if three:
            data_mix, smr = buf['speech_music'].take(batchSize, lo, hi)
            parts.append(data_mix)
        host_batch = isinstance(parts[0], np.ndarray)
        if host_batch:
            batchData = np.concatenate(parts, axis=0)  # this rank's rows [lo, hi) of every class block
            if 'Lemaire_et_al' in PARAMS['Model'] and per_file:
                batchData = np.transpose(batchData, axes=(0, 2, 1))  # per-file callables return (nP, F, W)
            # host batches draw their noise from numpy's global state: the draw has the GLOBAL batch's shape on every rank (the
            # ranks consume the same random numbers and their states stay in step) and a rank adds its rows of it
            if PARAMS['data_augmentation_with_noise']:
                if mine is None:
                    batchData = batching.noise_augmentation(batchData, rng)
                else:
                    full = np.zeros((len(classes_) * int(batchSize),) + batchData.shape[1:], batchData.dtype)
                    full[mine] = batchData
                    batchData = batching.noise_augmentation(full, rng)[mine]
        else:
            import torch
            batchData = torch.cat(parts, dim=0)  # device patches are already (N, W, F)
            if PARAMS['data_augmentation_with_noise']:  # the scale from numpy (same on all ranks), the noise from torch's generator
                batchData = batching.noise_augmentation(batchData, rng)
        if three:
            lab = batching.make_labels_3class(batchSize, np.asarray(smr, dtype=np.float64))
        else:  # two classes: only music / speech rows
            lab = batching.make_labels_3class(batchSize, np.zeros(batchSize))
            lab = {k: v[:2 * batchSize] for k, v in lab.items()}
            lab['3C'] = to_categorical([0] * batchSize + [1] * batchSize, 2)
        if mine is not None:
            lab = {k: v[mine] for k, v in lab.items()}
            if dist is not None and check_every and batch_count % int(check_every) == 0:
                _check_same_state(dist, batch_count)
        batch_count += 1
        if 'MTL' in PARAMS['Model']:
            yield batchData, lab
        else:
            yield batchData, lab['3C']


def test_file_wise_generator(PARAMS, file_name_sp, file_name_mu, target_dB, featuregram_fn=None, patches_fn=None):
    """All patches of ONE test file + their one-hot labels (Proposed_Work_Results.py:459-496).  The patch shift is the
    reference's hard-coded 68, not PARAMS['W_shift'].  Device path: float32 tensor (nP, W, 2F); with the per-file callables
    injected: what they return, transposed to (nP, W, F) for the TCN models."""
    n_fft = PARAMS['n_fft'][PARAMS['Model']]
    n_mels = PARAMS['n_mels'][PARAMS['Model']]
    featName = PARAMS['featName'][PARAMS['Model']]
    if PARAMS.get('frame_level_scaling') or PARAMS.get('skewness_vector'):
        raise ValueError("frame_level_scaling / skewness_vector are not wired into this path")
    if file_name_mu == '':
        spec, label = ('speech', file_name_sp, '', None), 1
    elif file_name_sp == '':
        spec, label = ('music', '', file_name_mu, None), 0
    else:
        spec, label = ('speech_music', file_name_sp, file_name_mu, target_dB), 2
    if featuregram_fn is not None or patches_fn is not None:
        from .lib import preprocessing as pp
        fv = (featuregram_fn or pp.get_featuregram)(PARAMS, spec[0], PARAMS['feature_opDir'], spec[1], spec[2], spec[3], n_fft,
                                                     n_mels, featName, save_feat=False)
        batchData = (patches_fn or pp.get_feature_patches)(PARAMS, fv, PARAMS['W'], HARDCODED_TEST_SHIFT, featName)
        if 'Lemaire_et_al' in PARAMS['Model']:
            batchData = np.transpose(batchData, axes=(0, 2, 1))
    else:
        saved = PARAMS.get('save_features', True)
        PARAMS['save_features'] = False  # save_feat=False at :465-469
        try:
            batchData = _device_patches_for(PARAMS, [spec], featName, n_fft, n_mels, PARAMS['W'], HARDCODED_TEST_SHIFT)[0]
        finally:
            PARAMS['save_features'] = saved
    numLab = int(batchData.shape[0])
    return batchData, to_categorical([label] * numLab, num_classes=len(PARAMS['classes']))
