"""Conv2D MTL baselines: host wrapper over `smh_cnn_*` (include/smh.h).

  get_Doukhan_MTL_model      lib/proposed_architectures.py:425-511
  get_Papakostas_MTL_model   lib/proposed_architectures.py:516-588
  get_Jang_MTL_model         lib/proposed_architectures.py:650-764

The layer graph, the parameter table (names, Keras shapes, order) and all arithmetic live in libsmh.so
(csrc/smh_cnn.hip); this class only keeps the host copy of the weights, initialises them the way the reference's
initialisers do, and moves tensors.  Training (`fit` / `train_on_batch` / `evaluate`, cnn_training.py over
`smh_cnn_train_step_f32`) is built for all three.
"""
from __future__ import annotations

import ctypes as C
import json
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .persistence import ModelSurfaceMixin
from .cnn_training import CnnTrainingMixin
from .model import head_spec

KINDS = {"Doukhan": 0, "Papakostas": 1, "Jang": 2}
# initial learning rates returned next to the model (proposed_architectures.py:499, 574, 751)
LEARNING_RATE = {"Doukhan": 0.0001, "Papakostas": 0.001, "Jang": 0.001}


class CnnMTL(CnnTrainingMixin, ModelSurfaceMixin):
    """`model` object of get_{Doukhan,Papakostas,Jang}_MTL_model."""

    def __init__(self, kind, input_shape, n_classes=3, seed=None, n_mels=120, n_fft=512, fs=16000, fc_width=0,
                 loss_weights=None):
        if kind not in KINDS:
            raise ValueError("kind must be one of %s" % sorted(KINDS))
        self.lib = _lib.require_gpu()
        self.kind, self.n_classes = kind, int(n_classes)
        self.in_h, self.in_w = int(input_shape[0]), int(input_shape[1])
        if len(input_shape) > 2 and int(input_shape[2]) != 1:
            raise ValueError("input_shape must be (H, W, 1), got %s" % (tuple(input_shape),))
        self.n_mels, self.n_fft, self.fs, self.fc_width = int(n_mels), int(n_fft), float(fs), int(fc_width)
        cfg = _lib.CnnCfg(KINDS[kind], self.in_h, self.in_w, self.n_classes, self.n_mels, self.n_fft, int(fc_width),
                          self.fs)
        h = C.c_void_p()
        _lib.check(self.lib.smh_cnn_create(C.byref(cfg), C.byref(h)), "smh_cnn_create")
        self._h = h
        self.out_dim = self.lib.smh_cnn_out_dim(self._h)
        self.feat_dim = self.lib.smh_cnn_feat_dim(self._h)
        self.initial_learning_rate = LEARNING_RATE[kind]
        self._spec = []  # (name, shape, offset)
        name = C.create_string_buffer(96)
        shape, nd, off = (C.c_int * 4)(), C.c_int(), C.c_size_t()
        for i in range(self.lib.smh_cnn_num_tensors(self._h)):
            _lib.check(self.lib.smh_cnn_tensor_info(self._h, i, name, 96, shape, C.byref(nd), C.byref(off)),
                       "smh_cnn_tensor_info")
            self._spec.append((name.value.decode(), tuple(shape[:nd.value]), int(off.value)))
        self.weights = OrderedDict()
        self._init_weights(np.random.default_rng(seed))
        assert self.count_params() == self.lib.smh_cnn_num_params(self._h)
        self._dirty = True
        self._device_newer = False
        self.loss_weights = loss_weights
        self._init_training_state()

    def __del__(self):
        tr = getattr(self, "_trainer", None)
        if tr:
            self.lib.smh_cnn_trainer_destroy(tr)
            self._trainer = None
        h = getattr(self, "_h", None)
        if h:
            self.lib.smh_cnn_destroy(h)
            self._h = None

    # ---- initialisers of the reference ---------------------------------------------------------------------------
    def _init_weights(self, rng):
        mel = None
        for name, shape, _ in self._spec:
            leaf = name.rsplit("/", 1)[1]
            if leaf == "kernel":
                if "melCl" in name:  # Constant(get_kernel_initializer(...)): mel weights over time and 3 channels
                    if mel is None:
                        mel = self._mel_basis()
                    i = int(name.split("melCl")[1].split("/")[0])
                    nz = np.where(mel[i] > 0)[0]
                    k = np.repeat(mel[i, nz[0]:nz[-1] + 1][:, None], shape[1], axis=1)[:, :, None, None]
                    w = np.repeat(k, shape[3], axis=3).astype(np.float32)
                    assert w.shape == shape, (name, w.shape, shape)
                elif self.kind == "Papakostas" and not self._is_head(name):  # RandomNormal(stddev=0.01)
                    w = rng.normal(0.0, 0.01, size=shape).astype(np.float32)
                else:  # glorot_uniform == VarianceScaling(1, 'fan_avg', 'uniform')
                    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
                    fan_in, fan_out = rf * shape[-2], rf * shape[-1]
                    lim = np.sqrt(6.0 / (fan_in + fan_out))
                    w = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif leaf == "bias":
                fill = 0.1 if (self.kind == "Papakostas" and not self._is_head(name)) else 0.0  # Constant(0.1)
                w = np.full(shape, fill, np.float32)
            elif leaf in ("gamma", "moving_variance"):
                w = np.ones(shape, np.float32)
            else:  # beta, moving_mean
                w = np.zeros(shape, np.float32)
            self.weights[name] = w

    @staticmethod
    def _is_head(name):
        return name.split("/")[0] in ("S", "M", "N", "R")

    def _mel_basis(self):
        from .frontend import Frontend, FrontendConfig
        fe = Frontend(FrontendConfig(n_fft=self.n_fft, win_length=min(400, self.n_fft), n_mels=self.n_mels,
                                     mel_sr=self.fs))
        return fe.mel_basis()

    # ---- Keras-style surface -------------------------------------------------------------------------------------
    @property
    def output_names(self):
        return [n for n, _, _ in head_spec(self.n_classes)] + ["3C"]

    @property
    def metrics_names(self):
        return ["loss"] + [n + "_loss" for n in self.output_names] + ["3C_accuracy"]

    @property
    def input_shape(self):
        return (None, self.in_h, self.in_w, 1)

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, s, _ in self._spec))

    def weight_names(self):
        return [n for n, _, _ in self._spec]

    def _pull_weights(self):
        """After training steps the device copy is the master: refresh the host dict from it."""
        if self._device_newer:
            flat = np.empty(self.count_params(), np.float32)
            _lib.check(self.lib.smh_cnn_get_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size,
                                                    _lib.current_stream()),
                       "smh_cnn_get_weights")
            for name, shape, off in self._spec:
                self.weights[name] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()
            self._device_newer = False

    def get_weights(self):
        self._pull_weights()
        return [self.weights[n].copy() for n, _, _ in self._spec]

    def get_weights_dict(self):
        self._pull_weights()
        return self.weights

    def set_weights(self, arrays):
        arrays = list(arrays)
        if len(arrays) != len(self._spec):
            raise ValueError("set_weights: expected %d arrays, got %d" % (len(self._spec), len(arrays)))
        for (name, shape, _), a in zip(self._spec, arrays):
            a = np.asarray(a, dtype=np.float32)
            if a.shape != tuple(shape):
                raise ValueError("set_weights: %s expects shape %s, got %s" % (name, shape, a.shape))
            self.weights[name] = a.copy()
        self._dirty = True
        self._device_newer = False

    def set_weights_dict(self, d):
        self.set_weights([d[n] for n, _, _ in self._spec])

    def save_weights(self, path):
        """`.h5` / `.hdf5`: HDF5 in Keras' weight-file layout (persistence.py); otherwise `<path>.npz`."""
        from .persistence import save_weights_file
        self._pull_weights()
        return save_weights_file(path, self.weights)

    def load_weights(self, path):
        from .persistence import load_weights_file
        self.set_weights_dict(load_weights_file(path))

    def to_json(self):
        return json.dumps({"class_name": self.kind + "_MTL", "config": {
            "input_shape": [self.in_h, self.in_w, 1], "n_classes": self.n_classes, "n_mels": self.n_mels,
            "n_fft": self.n_fft, "fs": self.fs, "fc_width": self.fc_width, "outputs": self.output_names}})

    def summary(self, print_fn=print):
        print_fn("Model: %s_MTL, input (None, %d, %d, 1)" % (self.kind, self.in_h, self.in_w))
        for name, shape, _ in self._spec:
            print_fn("  %-40s %-22s %d" % (name, str(tuple(shape)), int(np.prod(shape))))
        print_fn("Total params: %d" % self.count_params())

    # ---- inference -----------------------------------------------------------------------------------------------
    def _sync_weights(self):
        if self._dirty:
            flat = np.concatenate([self.weights[n].ravel() for n, _, _ in self._spec]).astype(np.float32)
            _lib.check(self.lib.smh_cnn_set_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size,
                                                    _lib.current_stream()),
                       "smh_cnn_set_weights")
            self._dirty = False

    def forward_device(self, x, out=None, features=None, dtype="f32"):
        """x: float32 CUDA tensor (N, H, W) or (N, H, W, 1) -> (N, out_dim) [S|M|(N)|R|3C] on the device.
        dtype="bf16": bf16 GEMM operands with f32 accumulation (smh_cnn_forward_bf16) -- faster, not the parity path."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            raise TypeError("forward_device expects a float32 CUDA tensor")
        if x.dim() == 4 and x.shape[3] == 1:
            x = x[..., 0]
        x = x.contiguous()
        if x.dim() != 3 or x.shape[1] != self.in_h or x.shape[2] != self.in_w:
            raise ValueError("expected input (N, %d, %d[, 1]), got %s" % (self.in_h, self.in_w, tuple(x.shape)))
        self._sync_weights()
        N = x.shape[0]
        if out is None:
            out = torch.empty((N, self.out_dim), dtype=torch.float32, device=x.device)
        nbytes = self.lib.smh_cnn_workspace_bytes(self._h, N)
        work = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=x.device)
        fn = self.lib.smh_cnn_forward_f32 if dtype == "f32" else self.lib.smh_cnn_forward_bf16
        _lib.check(fn(
            self._h, C.c_void_p(x.data_ptr()), N, C.c_void_p(out.data_ptr()),
            None if features is None else C.c_void_p(features.data_ptr()), C.c_void_p(work.data_ptr()), work.numel(),
            _lib.current_stream()), "smh_cnn_forward_" + dtype)
        return out

    def split_outputs(self, out):
        res, col = [], 0
        for _, odim, _ in head_spec(self.n_classes):
            res.append(out[:, col:col + odim])
            col += odim
        res.append(out[:, col:col + self.n_classes])
        return res

    def predict(self, x, batch_size=None, verbose=0, dtype="f32"):
        """model.predict(x=batchData) -> [S, M, (N,) R, 3C] numpy arrays (Proposed_Work_Results.py:520,586)."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        elif x.dtype != torch.float32:
            x = x.float()
        out = self.forward_device(x.cuda(), dtype=dtype)
        host = out.cpu().numpy()  # one copy for all outputs
        return [np.ascontiguousarray(o) for o in self.split_outputs(host)]
