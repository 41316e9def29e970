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

from . import _lib
from .training import HEAD_DROPOUT, TrainingMixin, _cur_stream


class CnnTrainingMixin(TrainingMixin):
    """Mixed into sm_hpss_mtl_amd.cnn_models.CnnMTL; fit / evaluate / pack_targets come from TrainingMixin."""

    def _init_training_state(self):
        self._trainer = None
        self._trainer_cap = 0
        self._grad_view = None
        self._drop_spec = []
        self.iterations = 0
        self.optimizer = "sgd" if self.kind == "Papakostas" else "adam"
        self.beta_1, self.beta_2, self.epsilon, self.momentum = 0.9, 0.999, 1e-7, 0.0
        self.decay_steps, self.decay_rate = 700, 0.1
        self._rng = torch.Generator(device="cuda")
        self._rng.manual_seed(1234)

    def learning_rate(self, step=None):
        step = self.iterations if step is None else step
        if self.kind == "Papakostas":  # ExponentialDecay(0.001, decay_steps=700, decay_rate=0.1), not staircase
            return self.initial_learning_rate * self.decay_rate ** (step / float(self.decay_steps))
        return self.initial_learning_rate

    def _get_trainer(self, n):
        if self._trainer is None or n > self._trainer_cap:
            if self._trainer is not None:
                self.lib.smh_cnn_trainer_destroy(self._trainer)
                self._trainer = None
            cap = max(n, 48)
            h = C.c_void_p()
            _lib.check(self.lib.smh_cnn_trainer_create(self._h, cap, C.byref(h)), "smh_cnn_trainer_create")
            self._trainer, self._trainer_cap, self._grad_view = h, cap, None
            self._drop_spec = []
            dim, rate = C.c_size_t(), C.c_float()
            for i in range(self.lib.smh_cnn_trainer_num_dropouts(h)):
                _lib.check(self.lib.smh_cnn_trainer_dropout_info(h, i, C.byref(dim), C.byref(rate)), "smh_cnn_trainer_dropout_info")
                self._drop_spec.append((int(dim.value), float(rate.value)))
        return self._trainer

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

    def _grad_tensor(self):
        if self._grad_view is None:
            ptr = self.lib.smh_cnn_trainer_grad_ptr(self._trainer)
            n = self.count_params()

            class _Holder:
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            self._grad_view = torch.as_tensor(_Holder(), device="cuda")
        return self._grad_view

    def gradients(self):
        """dict name -> gradient of the last train_on_batch(apply=False) (before grad_scale and the l2 term)."""
        flat = self._grad_tensor().cpu().numpy()
        return {name: flat[off:off + int(np.prod(shape))].reshape(shape).copy() for name, shape, off in self._spec}

    def train_on_batch(self, x, y, drop="auto", drop_heads="auto", apply=True):
        """One optimiser step.  Returns [loss, <per-output losses>, 3C_accuracy] like Keras.
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
        losses = torch.empty(n_heads + 4, dtype=torch.float32, device="cuda")
        p = lambda t: None if t is None else C.c_void_p(t.contiguous().data_ptr())  # noqa: E731
        _lib.check(self.lib.smh_cnn_train_step_f32(tr, p(x), p(yt), n, p(drop), p(drop_heads), self._loss_weight_array(),
                                                   p(losses), _cur_stream()), "smh_cnn_train_step_f32")
        if apply:
            self.apply_gradients()
        lv = losses.cpu().numpy()
        reg = float(lv[n_heads + 3])
        return [float(lv[n_heads + 1] + reg)] + [float(v) for v in lv[: n_heads + 1]] + [float(lv[n_heads + 2])]

    def apply_gradients(self):
        """All-reduce (if torch.distributed is initialised), then the optimiser update on the device."""
        import torch.distributed as dist
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._grad_tensor(), op=dist.ReduceOp.SUM)  # one flat bucket over RCCL
            scale = 1.0 / dist.get_world_size()
        adam = self.optimizer == "adam"
        _lib.check(self.lib.smh_cnn_trainer_apply_f32(self._trainer, 1 if adam else 0, self.learning_rate(),
                                                      self.beta_1 if adam else self.momentum, self.beta_2, self.epsilon, scale,
                                                      _cur_stream()), "smh_cnn_trainer_apply_f32")
        self.iterations += 1
        self._device_newer = True
