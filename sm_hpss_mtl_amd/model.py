"""B3_MTL host object: a Keras-style facade (`predict`, `get_weights`, `save_weights`, `to_json`, ...) over
the `smh_model` C ABI.  Mirrors what the reference's drivers call on the object returned by
`get_Lemaire_MTL_model` (lib/proposed_architectures.py:85-170; Proposed_Work_Results.py:345-374,520,586).

Weights are kept on the host as float32 numpy arrays in CANONICAL order (Keras array layouts):
  tcn/initial_conv kernel (1,F,32), bias (32)
  per (stack s, dilation d): conv kernel (3,32,32), bias, conv1x1 kernel (1,32,32), bias
  3C kernel (T*32, n_classes), bias
  per head (S, M, [N,] R): dense kernel (T*32,16), bias, BN gamma, beta, moving_mean, moving_variance,
                           out kernel (16, odim), out bias
and uploaded (re-packed into MFMA operand order by libsmh) whenever they change.
"""
from __future__ import annotations

import ctypes as C
import json
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .persistence import ModelSurfaceMixin
from .training import TrainingMixin


def head_spec(n_classes: int):
    """(name, out_dim, activation) of the auxiliary heads in Keras output order
    (proposed_architectures.py:25-80,154; 5_class_classification.py:150-215,286)."""
    if n_classes == 5:
        return [("S", 1, "sigmoid"), ("M", 1, "sigmoid"), ("N", 1, "sigmoid"), ("R", 3, "linear")]
    return [("S", 1, "sigmoid"), ("M", 1, "sigmoid"), ("R", 2, "linear")]


def weight_spec(n_feat, patch_size, n_classes, nb_filters=32, kernel_size=3, nb_stacks=3, n_dil=8, block_variant=0):
    """Ordered (name, shape, fan_in, fan_out|None) in canonical order; fan_out None -> not glorot.
    block_variant 0: the keras-tcn 2.3.x block; 1: the two-convolution block of keras-tcn >= 2.8 (include/smh.h)."""
    Cf, D = nb_filters, patch_size * nb_filters
    spec = []
    if block_variant == 0:
        spec = [("tcn/initial_conv/kernel", (1, n_feat, Cf), n_feat, Cf), ("tcn/initial_conv/bias", (Cf,), 0, None)]
    cin = n_feat
    for s in range(nb_stacks):
        for i in range(n_dil):
            p = "tcn/s%d_d%d" % (s, 2 ** i)
            if block_variant == 0:
                spec += [(p + "/conv/kernel", (kernel_size, Cf, Cf), kernel_size * Cf, kernel_size * Cf),
                         (p + "/conv/bias", (Cf,), 0, None),
                         (p + "/conv1x1/kernel", (1, Cf, Cf), Cf, Cf), (p + "/conv1x1/bias", (Cf,), 0, None)]
            else:
                spec += [(p + "/conv0/kernel", (kernel_size, cin, Cf), kernel_size * cin, kernel_size * Cf), (p + "/conv0/bias", (Cf,), 0, None),
                         (p + "/conv1/kernel", (kernel_size, Cf, Cf), kernel_size * Cf, kernel_size * Cf), (p + "/conv1/bias", (Cf,), 0, None)]
                if cin != Cf:
                    spec += [(p + "/matching/kernel", (1, cin, Cf), cin, Cf), (p + "/matching/bias", (Cf,), 0, None)]
                cin = Cf
    spec += [("3C/kernel", (D, n_classes), D, n_classes), ("3C/bias", (n_classes,), 0, None)]
    for name, odim, _ in head_spec(n_classes):
        spec += [(name + "/dense/kernel", (D, 16), D, 16), (name + "/dense/bias", (16,), 0, None),
                 (name + "/bn/gamma", (16,), 1, None), (name + "/bn/beta", (16,), 0, None),
                 (name + "/bn/moving_mean", (16,), 0, None), (name + "/bn/moving_variance", (16,), 1, None),
                 (name + "/out/kernel", (16, odim), 16, odim), (name + "/out/bias", (odim,), 0, None)]
    return spec


def initial_weights(n_feat=240, patch_size=68, n_classes=3, seed=None, nb_filters=32, kernel_size=3, nb_stacks=3,
                    n_dilations=8, block_variant=0):
    """(dropout_rate, OrderedDict name -> float32 array) of a freshly built model: Keras defaults (glorot_uniform
    kernels, zero biases, BatchNormalization gamma = moving_variance = 1) and the build-time draw of the spatial
    dropout rate (proposed_architectures.py:136).  Host-only (numpy): `B3MTL.__init__` and the generator of
    tests/golden/bench_golden.npz both call it, so `B3MTL(seed=s)` is reproducible without a GPU."""
    rng = np.random.default_rng(seed)
    dropout_rate = float(rng.uniform(0.05, 0.5))
    weights = OrderedDict()
    for name, shape, fan_in, fan_out in weight_spec(n_feat, patch_size, n_classes, nb_filters, kernel_size, nb_stacks,
                                                    n_dilations, block_variant):
        if fan_out is not None:
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            weights[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        else:
            weights[name] = np.full(shape, float(fan_in), np.float32)
    return dropout_rate, weights


class B3MTL(TrainingMixin, ModelSurfaceMixin):
    """`model` object of get_Lemaire_MTL_model.  Inference runs entirely in libsmh (HIP)."""

    def __init__(self, n_feat=240, patch_size=68, n_classes=3, TR_STEPS=1, loss_weights=None, seed=None,
                 nb_filters=32, kernel_size=3, nb_stacks=3, n_dilations=8, tcn_block="2.3"):
        """tcn_block: residual block of the third-party `tcn.TCN` the model was built with -- "2.3" (keras-tcn 2.3.x, what the
        reference's call binds under; default) or "2.8" (the two-convolution block of later releases; inference only)."""
        if str(tcn_block) not in ("2.3", "2.8"):
            raise ValueError("tcn_block must be '2.3' or '2.8', got %r" % (tcn_block,))
        self.tcn_block = str(tcn_block)
        self.block_variant = 0 if self.tcn_block == "2.3" else 1
        self.lib = _lib.require_gpu()
        self.n_feat, self.patch_size, self.n_classes = int(n_feat), int(patch_size), int(n_classes)
        self.nb_filters, self.kernel_size, self.nb_stacks, self.n_dilations = nb_filters, kernel_size, nb_stacks, n_dilations
        self.TR_STEPS, self.loss_weights = TR_STEPS, loss_weights
        # proposed_architectures.py:136 draws the (training-only) spatial dropout rate at build time
        self.dropout_rate, self.weights = initial_weights(self.n_feat, self.patch_size, self.n_classes, seed, nb_filters,
                                                          kernel_size, nb_stacks, n_dilations, self.block_variant)
        self.initial_learning_rate = 0.002
        cfg = _lib.ModelCfg(self.n_feat, self.patch_size, self.n_classes, nb_filters, kernel_size, nb_stacks, n_dilations,
                            self.block_variant)
        h = C.c_void_p()
        _lib.check(self.lib.smh_model_create(C.byref(cfg), C.byref(h)), "smh_model_create")
        self._h = h
        self.out_dim = self.lib.smh_model_out_dim(self._h)
        self._spec = weight_spec(self.n_feat, self.patch_size, self.n_classes, nb_filters, kernel_size, nb_stacks, n_dilations,
                                 self.block_variant)
        assert self.count_params() == self.lib.smh_model_num_params(self._h)
        self._dirty = True          # host copy newer than the device master
        self._device_newer = False  # device master newer than the host copy (after optimiser steps)
        self._init_training_state()

    def __del__(self):
        t = getattr(self, "_trainer", None)
        if t:
            self.lib.smh_trainer_destroy(t)
            self._trainer = None
        h = getattr(self, "_h", None)
        if h:
            self.lib.smh_model_destroy(h)
            self._h = None

    # ---- Keras-style surface -----------------------------------------------------------------
    @property
    def output_names(self):
        return [n for n, _, _ in head_spec(self.n_classes)] + ["3C"]

    @property
    def metrics_names(self):
        """Proposed_Work_Results.py:887 expects ['loss','S_loss','M_loss','R_loss','3C_loss','3C_accuracy']."""
        return ["loss"] + [n + "_loss" for n in self.output_names] + ["3C_accuracy"]

    def count_params(self):
        return int(sum(int(np.prod(s)) for _, s, _, _ in self._spec))

    def _pull_weights(self):
        if self._device_newer:
            flat = np.empty(self.count_params(), np.float32)
            _lib.check(self.lib.smh_model_get_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size,
                                                      _lib.current_stream()),
                       "smh_model_get_weights")
            o = 0
            for name, shape, _, _ in self._spec:
                n = int(np.prod(shape))
                self.weights[name] = flat[o:o + n].reshape(shape).copy()
                o += n
            self._device_newer = False

    def get_weights(self):
        self._pull_weights()
        return [self.weights[n].copy() for n, _, _, _ in self._spec]

    def get_weights_dict(self):
        self._pull_weights()
        return self.weights

    def set_weights(self, arrays):
        arrays = list(arrays)
        if len(arrays) != len(self._spec):
            raise ValueError("set_weights: expected %d arrays, got %d" % (len(self._spec), len(arrays)))
        for (name, shape, _, _), a in zip(self._spec, arrays):
            a = np.asarray(a, dtype=np.float32)
            if a.shape != tuple(shape):
                raise ValueError("set_weights: %s expects shape %s, got %s" % (name, shape, a.shape))
            self.weights[name] = a.copy()
        self._dirty = True
        self._device_newer = False

    def set_weights_dict(self, d):
        self.set_weights([d[n] for n, _, _, _ in self._spec])

    def save_weights(self, path):
        """`.h5` / `.hdf5`: HDF5 in Keras' weight-file layout (persistence.py); otherwise `<path>.npz`."""
        from .persistence import save_weights_file
        self._pull_weights()
        return save_weights_file(path, self.weights)

    def load_weights(self, path, arch_json=None):
        """Weights written by `save_weights` (.h5 / .npz), or an .h5 file written by Keras itself for this architecture:
        its auto-generated layer names are mapped through the architecture JSON (`arch_json`: path or text; default
        `<path without .h5>.json`, the file the reference writes next to the weights)."""
        from .persistence import load_weights_file
        self.set_weights_dict(load_weights_file(path, arch_json=arch_json))

    def to_json(self):
        return json.dumps({"class_name": "B3_MTL", "config": {
            "n_feat": self.n_feat, "patch_size": self.patch_size, "n_classes": self.n_classes,
            "nb_filters": self.nb_filters, "kernel_size": self.kernel_size, "nb_stacks": self.nb_stacks,
            "n_dilations": self.n_dilations, "dropout_rate": self.dropout_rate, "outputs": self.output_names,
            "tcn_block": self.tcn_block}})

    def summary(self, print_fn=print):
        print_fn("Model: B3_MTL (Lemaire et al. TCN + MTL heads), input (None, %d, %d)" % (self.patch_size, self.n_feat))
        for name, shape, _, _ in self._spec:
            print_fn("  %-40s %-18s %d" % (name, str(tuple(shape)), int(np.prod(shape))))
        print_fn("Total params: %d" % self.count_params())

    # ---- inference -----------------------------------------------------------------------------
    def _sync_weights(self):
        if self._dirty:
            flat = np.concatenate([self.weights[n].ravel() for n, _, _, _ in self._spec]).astype(np.float32)
            _lib.check(self.lib.smh_model_set_weights(self._h, flat.ctypes.data_as(C.c_void_p), flat.size,
                                                      _lib.current_stream()),
                       "smh_model_set_weights")
            self._dirty = False

    def forward_device(self, x, out=None, trunk=None, dtype="f32"):
        """x: float32 CUDA tensor (N, W, n_feat) -> (N, out_dim) tensor [S|M|(N)|R|3C] on the device.
        dtype="bf16" selects the mixed-precision kernel (bf16 matrix-core operands, f32 everything else): faster,
        not the parity path."""
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
            raise TypeError("forward_device expects a float32 CUDA tensor")
        x = x.contiguous()
        if x.dim() != 3 or x.shape[1] != self.patch_size or x.shape[2] != self.n_feat:
            raise ValueError("expected input (N, %d, %d), got %s" % (self.patch_size, self.n_feat, tuple(x.shape)))
        self._sync_weights()
        N = x.shape[0]
        if out is None:
            out = torch.empty((N, self.out_dim), dtype=torch.float32, device=x.device)
        if dtype == "bf16":
            # "bf16" = SPLIT bf16 operands: every f32 operand travels as hi + lo (two bf16 values) and every product is three
            # bf16 MFMA (hi*hi + hi*lo + lo*hi) accumulated in f32 -- f32-grade arithmetic on the bf16 matrix pipe (7e-5 from the
            # f32 kernel), not an 8-bit-mantissa network.  Operands rounded to ONE bf16 land 3.5-5e-2 from f32, outside SURVEY
            # 8(d')'s 2e-2, and are not offered here (C ABI: smh_model_forward_bf16_ex(split = 0), measurement only).
            if trunk is not None:
                raise ValueError("the trunk tap is only available on the f32 path")
            _lib.check(self.lib.smh_model_forward_bf16_ex(
                self._h, C.c_void_p(x.data_ptr()), N, C.c_void_p(out.data_ptr()), 1,
                _lib.current_stream()), "smh_model_forward_bf16")
            return out
        if dtype != "f32":
            raise ValueError("dtype must be 'f32' or 'bf16' (split bf16 operands), got %r" % (dtype,))
        _lib.check(self.lib.smh_model_forward_f32(
            self._h, C.c_void_p(x.data_ptr()), N, C.c_void_p(out.data_ptr()),
            None if trunk is None else C.c_void_p(trunk.data_ptr()),
            _lib.current_stream()), "smh_model_forward_f32")
        return out

    def forward_from_x0(self, x0p, out=None, trunk=None, dtype="f32"):
        """Forward that starts from the per-half layer-0 partials (N, 2, W, 32) written by `Frontend.features_l0`.
        dtype as for `forward_device`; with "bf16" layer 0 stays exact f32 (it was computed by the feature kernel)."""
        if not (isinstance(x0p, torch.Tensor) and x0p.is_cuda and x0p.dtype == torch.float32):
            raise TypeError("forward_from_x0 expects a float32 CUDA tensor")
        x0p = x0p.contiguous()
        if self.block_variant != 0:
            raise ValueError("the layer-0 fusion exists for the keras-tcn 2.3.x block only")
        if x0p.dim() != 4 or tuple(x0p.shape[1:]) != (2, self.patch_size, 32):
            raise ValueError("expected (N, 2, %d, 32), got %s" % (self.patch_size, tuple(x0p.shape)))
        self._sync_weights()
        N = x0p.shape[0]
        if out is None:
            out = torch.empty((N, self.out_dim), dtype=torch.float32, device=x0p.device)
        if dtype == "bf16":
            if trunk is not None:
                raise ValueError("the trunk tap is only available on the f32 path")
            _lib.check(self.lib.smh_model_forward_x0_bf16(
                self._h, C.c_void_p(x0p.data_ptr()), N, C.c_void_p(out.data_ptr()), 1,
                _lib.current_stream()), "smh_model_forward_x0_bf16")
            return out
        if dtype != "f32":
            raise ValueError("dtype must be 'f32' or 'bf16' (split bf16 operands), got %r" % (dtype,))
        _lib.check(self.lib.smh_model_forward_x0_f32(
            self._h, C.c_void_p(x0p.data_ptr()), N, C.c_void_p(out.data_ptr()),
            None if trunk is None else C.c_void_p(trunk.data_ptr()),
            _lib.current_stream()), "smh_model_forward_x0_f32")
        return out

    def forward_dense(self, fv, shift=1, out=None):
        """Every hop-`shift` patch of a standardised featuregram batch fv (n_feat, Tc) through the network (dense file-level
        inference, DAFx12_Speech_Music_Detection_B3_MTL_v2.py:634-665) WITHOUT building the (nP, W, n_feat) patches: layer 0 once
        per frame, every patch a window of it (`smh_model_forward_dense_f32`).  Returns (nP, out_dim); nP = tools.extract_patches'
        count for Tc frames.  Needs Tc >= patch_size; shorter batches are tiled by get_feature_patches and go through forward_device."""
        if not (isinstance(fv, torch.Tensor) and fv.is_cuda and fv.dtype == torch.float32):
            raise TypeError("forward_dense expects a float32 CUDA tensor")
        if fv.dim() != 2 or fv.shape[0] != self.n_feat:
            raise ValueError("expected (%d, Tc), got %s" % (self.n_feat, tuple(fv.shape)))
        if self.block_variant != 0:
            raise ValueError("the layer-0 fusion exists for the keras-tcn 2.3.x block only")
        fv = fv.contiguous()
        Tc = int(fv.shape[1])
        self._sync_weights()
        nP = self.lib.smh_num_patches(Tc, self.patch_size, int(shift)) if Tc >= self.patch_size else -1
        if nP < 0:
            raise ValueError("forward_dense needs shift >= 1 and at least patch_size=%d frames, got Tc=%d shift=%d" % (self.patch_size, Tc, shift))
        if out is None:
            out = torch.empty((nP, self.out_dim), dtype=torch.float32, device=fv.device)
        elif tuple(out.shape) != (nP, self.out_dim):
            raise ValueError("out must be (%d, %d)" % (nP, self.out_dim))
        if nP == 0:
            return out
        nbytes = self.lib.smh_model_dense_workspace_bytes(self._h, Tc)
        work = torch.empty((nbytes // 4,), dtype=torch.float32, device=fv.device)
        got = _lib.check(self.lib.smh_model_forward_dense_f32(
            self._h, C.c_void_p(fv.data_ptr()), Tc, int(shift), C.c_void_p(work.data_ptr()), nbytes, C.c_void_p(out.data_ptr()),
            _lib.current_stream()), "smh_model_forward_dense_f32")
        if got != nP:
            raise RuntimeError("smh_model_forward_dense_f32 produced %d patches, expected %d" % (got, nP))
        return out

    def split_outputs(self, out):
        """(N, out_dim) -> list in Keras output order [S, M, (N,) R, 3C]."""
        res, col = [], 0
        for _, odim, _ in head_spec(self.n_classes):
            res.append(out[:, col:col + odim])
            col += odim
        res.append(out[:, col:col + self.n_classes])
        return res

    def predict(self, x, batch_size=None, verbose=0, dtype="f32"):
        """model.predict(x=batchData) -> [S, M, (N,) R, 3C] numpy arrays (Proposed_Work_Results.py:520,586)."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
        elif x.dtype != torch.float32:
            x = x.float()
        out = self.forward_device(x.cuda(), dtype=dtype)
        self.check_status()  # the forward is stream-ordered: a device-side give-up must become an exception, not a result
        host = out.cpu().numpy()  # ONE copy for all outputs (a copy per output is a host synchronisation per output)
        return [np.ascontiguousarray(o) for o in self.split_outputs(host)]

    def check_status(self):
        """Wait for the current stream and raise RuntimeError if a forward kernel recorded in the model's device error word
        that its outputs are not results (include/smh.h: smh_model_status).  `forward_device` / `forward_from_x0` only enqueue
        work; callers that keep results on the device call this before trusting them (`predict` and bench.py do)."""
        _lib.check(self.lib.smh_model_status(self._h, _lib.current_stream()), "smh_model_status")
