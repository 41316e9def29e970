"""Keras-style training surface of B3MTL: `train_on_batch`, `fit`, `evaluate`.

Mirrors how the reference drives the model (Proposed_Work_Results.py:275-312, 678-700):
  model.fit(generator, steps_per_epoch, epochs, validation_data=generator, validation_steps,
            callbacks=[CSVLogger, EarlyStopping(val_loss, min_delta=.01, patience=5, restore_best_weights),
                       ModelCheckpoint(save_best_only, save_weights_only)])
  model.evaluate(generator, steps) -> [loss, S_loss, M_loss, R_loss, 3C_loss, 3C_accuracy]
with the optimiser of lib/proposed_architectures.py:156-158: SGD(ExponentialDecay(0.002, 3*TR_STEPS, 0.1),
momentum=0.9, clipnorm=1).  All arithmetic runs in libsmh (HIP); torch supplies device memory, random
dropout masks and -- for data parallel training -- ONE all-reduce of the flat gradient over RCCL.
"""
from __future__ import annotations

import ctypes as C
import csv
import time

import numpy as np
import torch

from . import _lib

HEAD_DROPOUT = 0.4  # Dropout(0.4) of MTL_modifications (proposed_architectures.py:49,63,76)
L2 = 0.01


def _cur_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class History:
    def __init__(self):
        self.history = {}
        self.epoch = []


class TrainingMixin:
    """Mixed into sm_hpss_mtl_amd.model.B3MTL."""

    # ---- optimiser state --------------------------------------------------------------------
    def _init_training_state(self):
        self._trainer = None
        self._trainer_cap = 0
        self.iterations = 0
        self.momentum, self.clipnorm = 0.9, 1.0
        self.decay_steps = 3 * max(int(self.TR_STEPS), 1)
        self.decay_rate = 0.1
        self._rng = torch.Generator(device="cuda")
        self._rng.manual_seed(1234)

    def learning_rate(self, step=None):
        """ExponentialDecay(0.002, decay_steps=3*TR_STEPS, decay_rate=0.1), not staircase."""
        step = self.iterations if step is None else step
        return self.initial_learning_rate * self.decay_rate ** (step / float(self.decay_steps))

    def _get_trainer(self, n):
        if self._trainer is None or n > self._trainer_cap:
            if self._trainer is not None:
                self.lib.smh_trainer_destroy(self._trainer)
            cap = max(n, 64)
            h = C.c_void_p()
            _lib.check(self.lib.smh_trainer_create(self._h, cap, C.byref(h)), "smh_trainer_create")
            self._trainer, self._trainer_cap = h, cap
            self._grad_view = None
        return self._trainer

    def _grad_tensor(self):
        """torch view of the trainer's flat gradient (for the RCCL all-reduce)."""
        if self._grad_view is None:
            ptr = self.lib.smh_trainer_grad_ptr(self._trainer)
            n = self.count_params()

            class _Holder:
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            self._grad_view = torch.as_tensor(_Holder(), device="cuda")
        return self._grad_view

    # ---- targets ----------------------------------------------------------------------------------
    def pack_targets(self, y):
        """dict / list in Keras output order [S, M, (N,) R, 3C] -> (N, out_dim) float32 CUDA tensor."""
        names = self.output_names
        if isinstance(y, dict):
            y = [y[k] for k in names]
        cols = []
        for a in y:
            a = torch.as_tensor(np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float32)
            cols.append(a.reshape(a.shape[0], -1))
        t = torch.cat(cols, dim=1)
        if t.shape[1] != self.out_dim:
            raise ValueError("targets have %d columns, model outputs %d" % (t.shape[1], self.out_dim))
        return t.cuda().contiguous()

    def _loss_weight_array(self):
        w = [1.0] * len(self.output_names)
        if self.loss_weights:
            for i, n in enumerate(self.output_names):
                w[i] = float(self.loss_weights.get(n, 1.0))
        return (C.c_float * len(w))(*w)

    # ---- one step ---------------------------------------------------------------------------------
    def train_on_batch(self, x, y, drop_tcn="auto", drop_heads="auto", apply=True):
        """One optimiser step.  Returns [loss, <per-output losses>, 3C_accuracy] like Keras.
        drop_*: "auto" draws masks with the model's rates, None disables dropout, or pass mask tensors."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(device="cuda", dtype=torch.float32).contiguous()
        n = x.shape[0]
        if x.dim() != 3 or x.shape[1] != self.patch_size or x.shape[2] != self.n_feat:
            raise ValueError("expected input (N, %d, %d), got %s" % (self.patch_size, self.n_feat, tuple(x.shape)))
        yt = y if (isinstance(y, torch.Tensor) and y.is_cuda and y.dim() == 2) else self.pack_targets(y)
        self._sync_weights()
        tr = self._get_trainer(n)
        n_blocks, n_heads = self.nb_stacks * self.n_dilations, len(self.output_names) - 1
        if isinstance(drop_tcn, str):
            keep = 1.0 - self.dropout_rate
            drop_tcn = (torch.rand((n, n_blocks, 32), device="cuda", generator=self._rng) < keep).float() / keep
        if isinstance(drop_heads, str):
            keep = 1.0 - HEAD_DROPOUT
            drop_heads = (torch.rand((n, n_heads, 16), device="cuda", generator=self._rng) < keep).float() / keep
        losses = torch.empty(n_heads + 4, dtype=torch.float32, device="cuda")
        p = lambda t: None if t is None else C.c_void_p(t.contiguous().data_ptr())  # noqa: E731
        _lib.check(self.lib.smh_train_step_f32(tr, p(x), p(yt), n, p(drop_tcn), p(drop_heads), self._loss_weight_array(),
                                               p(losses), _cur_stream()), "smh_train_step_f32")
        if apply:
            self.apply_gradients()
        lv = losses.cpu().numpy()
        reg = float(lv[n_heads + 3])  # l2 penalty of the weights this step ran with, computed on the device
        # Keras order: total loss, one loss per output, then the metric
        return [float(lv[n_heads + 1] + reg)] + [float(v) for v in lv[: n_heads + 1]] + [float(lv[n_heads + 2])]

    def apply_gradients(self):
        """All-reduce (if torch.distributed is initialised), clip, momentum update, repack."""
        import torch.distributed as dist
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._grad_tensor(), op=dist.ReduceOp.SUM)  # ONE flat bucket (0.9 MB) over RCCL
            scale = 1.0 / dist.get_world_size()
        _lib.check(self.lib.smh_trainer_apply_sgd_f32(self._trainer, self.learning_rate(), self.momentum, self.clipnorm, scale,
                                                      _cur_stream()), "smh_trainer_apply_sgd_f32")
        self.iterations += 1
        self._device_newer = True

    def _l2_penalty(self):
        w = self.get_weights_dict()
        return float(sum(L2 * np.sum(w[n + "/dense/kernel"].astype(np.float64) ** 2) for n in self.output_names[:-1]))

    # ---- evaluate ---------------------------------------------------------------------------------
    def _losses_inference(self, x, y):
        """Losses / accuracy of one batch in inference mode (Keras `evaluate` semantics)."""
        outs = self.predict(x)
        yl = y if isinstance(y, (list, tuple)) else [y[k] for k in self.output_names]
        eps = 1e-7
        per = []
        for name, o, t in zip(self.output_names, outs, yl):
            t = np.asarray(t, np.float64).reshape(o.shape)
            o = o.astype(np.float64)
            if name == "3C":
                per.append(float(np.mean(-np.sum(t * np.log(np.clip(o / o.sum(1, keepdims=True), eps, 1 - eps)), axis=1))))
            elif name == "R":
                per.append(float(np.mean((o - t) ** 2)))
            else:
                oc = np.clip(o, eps, 1 - eps)
                per.append(float(np.mean(-(t * np.log(oc + eps) + (1 - t) * np.log(1 - oc + eps)))))
        lw = [float((self.loss_weights or {}).get(n, 1.0)) for n in self.output_names]
        acc = float(np.mean(outs[-1].argmax(1) == np.asarray(yl[-1]).argmax(1)))
        return [sum(a * b for a, b in zip(lw, per)) + self._l2_penalty()] + per + [acc]

    def evaluate(self, x=None, y=None, steps=None, verbose=0, **_):
        """model.evaluate(generator, steps) or evaluate(x, y) -> list matching `metrics_names`."""
        if y is not None:
            return self._losses_inference(x, y)
        tot, cnt = None, 0
        for _ in range(int(steps)):
            bx, by = next(x)
            v = np.array(self._losses_inference(bx, by))
            tot = v if tot is None else tot + v
            cnt += 1
        return list(tot / max(cnt, 1))

    # ---- fit ------------------------------------------------------------------------------------------
    def fit(self, x=None, y=None, steps_per_epoch=None, epochs=1, validation_data=None, validation_steps=None,
            verbose=1, csv_log=None, checkpoint_path=None, early_stopping=None, batch_size=None, **_):
        """Generator- or array-driven training loop.
        early_stopping: dict(monitor='val_loss', min_delta=0.01, patience=5, restore_best_weights=True)
        checkpoint_path: best-`val_loss` weights are saved there (ModelCheckpoint(save_best_only, weights only))
        csv_log: per-epoch CSV like keras.callbacks.CSVLogger."""
        hist = History()
        names = self.metrics_names
        es = dict(monitor="val_loss", min_delta=0.0, patience=None, restore_best_weights=False)
        es.update(early_stopping or {})
        best, best_w, wait = np.inf, None, 0
        rows = []
        arrays = y is not None
        if arrays:
            xs = np.asarray(x, np.float32)
            yl = y if isinstance(y, (list, tuple)) else [y[k] for k in self.output_names]
            bs = batch_size or 32
            steps_per_epoch = steps_per_epoch or int(np.ceil(len(xs) / bs))
        for ep in range(int(epochs)):
            t0 = time.time()
            agg = np.zeros(len(names))
            for s in range(int(steps_per_epoch)):
                if arrays:
                    sl = slice((s * bs) % len(xs), (s * bs) % len(xs) + bs)
                    bx, by = xs[sl], [np.asarray(a)[sl] for a in yl]
                else:
                    bx, by = next(x)
                agg += np.array(self.train_on_batch(bx, by))
            logs = {n: float(v) for n, v in zip(names, agg / max(int(steps_per_epoch), 1))}
            if validation_data is not None:
                if isinstance(validation_data, (tuple, list)) and not hasattr(validation_data, "__next__"):
                    val = self.evaluate(validation_data[0], validation_data[1])
                else:
                    val = self.evaluate(validation_data, steps=validation_steps)
                logs.update({"val_" + n: float(v) for n, v in zip(names, val)})
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            hist.epoch.append(ep)
            rows.append(dict(epoch=ep, **logs))
            if verbose:
                print("Epoch %d/%d - %.1fs - " % (ep + 1, epochs, time.time() - t0) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
            mon = logs.get(es["monitor"])
            if mon is not None:
                if mon < best - es["min_delta"]:
                    best, wait = mon, 0
                    if es["restore_best_weights"] or checkpoint_path:
                        best_w = self.get_weights()
                    if checkpoint_path:
                        self.save_weights(checkpoint_path)
                else:
                    wait += 1
                    if es["patience"] is not None and wait >= es["patience"]:
                        if verbose:
                            print("Early stopping at epoch %d" % (ep + 1))
                        break
        if es["restore_best_weights"] and best_w is not None:
            self.set_weights(best_w)
        if csv_log and rows:
            with open(csv_log, "w", newline="") as f:
                wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr.writeheader()
                wr.writerows(rows)
        return hist
