"""Keras-style training surface of the Conv2D MTL baselines: `train_on_batch`, `fit`, `evaluate`.

Optimisers as compiled by the reference: Doukhan Adam(1e-4) (lib/proposed_architectures.py:499-500), Papakostas
SGD(ExponentialDecay(1e-3, 700, 0.1)) (:572-574), Jang Adam(1e-3) (:750-751); Keras defaults beta_1 0.9, beta_2 0.999,
epsilon 1e-7.  The step itself -- training-mode forward, losses, backward, update, BatchNorm moving averages -- runs
in libsmh (smh_cnn_train.hip); torch supplies device memory, the random dropout masks and, for data-parallel
training, ONE all-reduce of the flat gradient over RCCL.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

import os

from . import _lib
from . import optimizers as _opt
from .training import HEAD_DROPOUT, TrainingMixin, _cur_stream


class CnnTrainingMixin(TrainingMixin):
    """Mixed into sm_hpss_mtl_amd.cnn_models.CnnMTL; compile / fit / evaluate / pack_targets come from TrainingMixin."""

    _TRAINER_API = ("smh_cnn_trainer_create", "smh_cnn_trainer_destroy", "smh_cnn_trainer_copy_state", "smh_cnn_trainer_grad_ptr",
                    "smh_cnn_trainer_bucket_floats")
    _MIN_TRAINER_CAP = 48

    def _init_training_state(self):
        self._trainer = None
        self._trainer_cap = 0
        self._grad_view = None
        self._drop_spec = []
        self.iterations = 0
        self.stop_training = False
        if self.kind == "Papakostas":  # lib/proposed_architectures.py:572-574
            self.optimizer = _opt.SGD(learning_rate=_opt.ExponentialDecay(self.initial_learning_rate, 700, 0.1))
        else:                           # :499-500 (Doukhan, 1e-4), :750-751 (Jang, 1e-3)
            self.optimizer = _opt.Adam(learning_rate=self.initial_learning_rate)
        self._rng = torch.Generator(device="cuda")
        self._rng.manual_seed(1234 + int(os.environ.get("RANK", "0")))

    def _set_optimizer(self, optimizer):
        if optimizer.kind not in ("sgd", "adam"):
            raise ValueError("the Conv2D MTL models train with SGD or Adam (as the reference compiles them), not %s" % optimizer.kind)
        if optimizer.clipnorm:
            raise ValueError("clipnorm is not part of the Conv2D MTL models' optimisers")
        self.optimizer = optimizer
        self.iterations = 0
        self._reset_optimizer_state()

    def _reset_optimizer_state(self):
        if self._trainer is not None:  # a new trainer starts from zeroed moments and step 0
            self.lib.smh_cnn_trainer_destroy(self._trainer)
            self._trainer, self._trainer_cap, self._grad_view = None, 0, None

    def _n_losses(self):
        return len(self.output_names) - 1 + 4

    def _on_new_trainer(self):
        self._drop_spec = []
        dim, rate = C.c_size_t(), C.c_float()
        for i in range(self.lib.smh_cnn_trainer_num_dropouts(self._trainer)):
            _lib.check(self.lib.smh_cnn_trainer_dropout_info(self._trainer, i, C.byref(dim), C.byref(rate)), "smh_cnn_trainer_dropout_info")
            self._drop_spec.append((int(dim.value), float(rate.value)))

    def _l2_penalty(self):
        """0.01 * sum w^2 over the kernels that carry kernel_regularizer=l2(): the heads' Dense(16) kernels; for Jang every
        mel-scale / Conv2D / Dense kernel and the '3C' kernel as well (proposed_architectures.py:630-747)."""
        w = self.get_weights_dict()
        names = [n + "/dense/kernel" for n in self.output_names[:-1]]
        if self.kind == "Jang":
            names += [k for k in w if k.endswith("/kernel") and (k.startswith(("conv", "fc", "3C")) or "_melCl" in k)]
        return float(sum(0.01 * np.sum(w[k].astype(np.float64) ** 2) for k in names))

    def dropout_spec(self, n=2):
        """[(dim, rate)] of the trunk's Dropout layers in graph order."""
        self._get_trainer(n)
        return list(self._drop_spec)

    def gradients(self):
        """dict name -> gradient of the last train_on_batch(apply=False) (before grad_scale and the l2 term)."""
        flat = self._grad_tensor().cpu().numpy()
        return {name: flat[off:off + int(np.prod(shape))].reshape(shape).copy() for name, shape, off in self._spec}

    def train_on_batch(self, x, y, drop="auto", drop_heads="auto", apply=True, sync=True):
        """One optimiser step.  Returns [loss, <per-output losses>, 3C_accuracy] like Keras (sync=False: the raw device
        tensor of the step's losses, see TrainingMixin.losses_to_list).
        drop: "auto" draws masks with the model's rates, None disables dropout, or a list of (N, dim_i) mask tensors
        (0 or 1/(1-rate_i)) in graph order; drop_heads likewise with one (N, n_heads, 16) tensor."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(device="cuda", dtype=torch.float32)
        if x.dim() == 4 and x.shape[3] == 1:
            x = x[..., 0]
        x = x.contiguous()
        n = x.shape[0]
        if x.dim() != 3 or x.shape[1] != self.in_h or x.shape[2] != self.in_w:
            raise ValueError("expected input (N, %d, %d[, 1]), got %s" % (self.in_h, self.in_w, tuple(x.shape)))
        yt = y if (isinstance(y, torch.Tensor) and y.is_cuda and y.dim() == 2) else self.pack_targets(y)
        self._sync_weights()
        tr = self._get_trainer(n)
        n_heads = len(self.output_names) - 1
        if isinstance(drop, str):
            drop = [(torch.rand((n, d), device="cuda", generator=self._rng) < 1.0 - r).float() / (1.0 - r)
                    for d, r in self._drop_spec]
        if drop is not None:
            if len(drop) != len(self._drop_spec):
                raise ValueError("expected %d dropout masks, got %d" % (len(self._drop_spec), len(drop)))
            parts = []
            for mk, (d, _) in zip(drop, self._drop_spec):
                mk = torch.as_tensor(mk, dtype=torch.float32).to("cuda").reshape(n, -1)
                if mk.shape[1] != d:
                    raise ValueError("dropout mask has %d columns, layer has %d" % (mk.shape[1], d))
                parts.append(mk.reshape(-1))
            drop = torch.cat(parts) if parts else None
        if isinstance(drop_heads, str):
            keep = 1.0 - HEAD_DROPOUT
            drop_heads = (torch.rand((n, n_heads, 16), device="cuda", generator=self._rng) < keep).float() / keep
        elif drop_heads is not None:
            drop_heads = torch.as_tensor(drop_heads, dtype=torch.float32).to("cuda")
        losses = torch.empty(self._n_losses(), dtype=torch.float32, device="cuda")
        p = lambda t: None if t is None else C.c_void_p(t.contiguous().data_ptr())  # noqa: E731
        _lib.check(self.lib.smh_cnn_train_step_f32(tr, p(x), p(yt), n, p(drop), p(drop_heads), self._loss_weight_array(),
                                                   p(losses), _cur_stream()), "smh_cnn_train_step_f32")
        if apply:
            self.apply_gradients()
        return self.losses_to_list(losses) if sync else losses

    def _apply_native(self, lr, scale, mask):
        """The optimiser update on the device (the all-reduce of the bucket happens in TrainingMixin.apply_gradients)."""
        o = self.optimizer
        adam = o.kind == "adam"
        _lib.check(self.lib.smh_cnn_trainer_apply_f32(self._trainer, 1 if adam else 0, lr, o.beta_1 if adam else o.momentum,
                                                      getattr(o, "beta_2", 0.999), getattr(o, "epsilon", 1e-7), scale,
                                                      _cur_stream()), "smh_cnn_trainer_apply_f32")
