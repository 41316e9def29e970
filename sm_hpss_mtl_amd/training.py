"""Keras-style training surface of B3MTL: `compile`, `train_on_batch`, `fit`, `evaluate`.

Mirrors how the reference drives the model (Proposed_Work_Results.py:275-312, 678-700):
  model.fit(generator, steps_per_epoch, epochs, validation_data=generator, validation_steps, verbose=1,
            callbacks=[CSVLogger(logFile), EarlyStopping(val_loss, min_delta=.01, patience=5, restore_best_weights),
                       ModelCheckpoint(weightFile, save_best_only, save_weights_only)])
  model.evaluate(generator, steps) -> [loss, S_loss, M_loss, R_loss, 3C_loss, 3C_accuracy]
with the optimiser of lib/proposed_architectures.py:156-158: SGD(ExponentialDecay(0.002, 3*TR_STEPS, 0.1),
momentum=0.9, clipnorm=1).  All arithmetic runs in libsmh (HIP); torch supplies device memory, random
dropout masks and -- for data parallel training -- ONE all-reduce of the bucket [flat gradient | BatchNorm batch
statistics] over RCCL.  Nothing in a step reads the device back: the losses of an epoch are summed on the device and
copied to the host once per epoch (`train_on_batch` returns host floats, as Keras does, unless sync=False).

Data parallel (torch.distributed initialised, world > 1).  Replicas start identical, every step applies the same
averaged gradient, so they stay identical -- and everything that DECIDES must see the same numbers on every rank, or one
rank leaves the loop while the others wait in the next all-reduce:
  * per step: one SUM all-reduce of the bucket (apply_gradients);
  * per epoch: the epoch's training losses and the validation losses are all-reduced (row-weighted mean over the ranks'
    shards = the loss of the global batch) and the resulting logs broadcast from rank 0, BEFORE the callbacks see them:
    EarlyStopping, ModelCheckpoint and restore-best take the same decision everywhere; a `stop_training` set on any rank
    (a user callback) stops all of them (MAX all-reduce);
  * rank 0 alone writes the checkpoint and the CSV log (`Callback.is_writer`);
  * the generator hands rank r the rows `shard_indices(3 * batchSize, r, world)` of each globally class-balanced batch
    (sm_hpss_mtl_amd.generators.generator), so class balance is a property of the global batch, not of a rank's draw.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

from . import _lib
from . import optimizers as _opt
from .callbacks import Callback, CSVLogger, EarlyStopping, ModelCheckpoint

HEAD_DROPOUT = 0.4  # Dropout(0.4) of MTL_modifications (proposed_architectures.py:49,63,76)
L2 = 0.01
TRAIN_TRUNK, TRAIN_3C, TRAIN_ALL = 1, 2, 0xFFFFFFFF
# fit() arguments of tf.keras that change nothing here (single process, generators consumed in the calling thread)
_FIT_IGNORED = ("workers", "use_multiprocessing", "max_queue_size", "shuffle", "class_weight", "sample_weight",
                "validation_freq", "validation_batch_size")
_LOSS_OF = {"S": "binary_crossentropy", "M": "binary_crossentropy", "N": "binary_crossentropy", "R": "mean_squared_error",
            "3C": "categorical_crossentropy"}


def _cur_stream():
    return _lib.current_stream()


def process_group():
    """torch.distributed when a process group with more than one rank is initialised, else None.  `SMH_DIST_SINGLE_RANK=1` also
    sends a ONE-rank group through every collective of the data-parallel path (gradient bucket, epoch logs, stop flag, generator
    state check): on a one-GPU machine that is the only way the RCCL backend itself gets executed (tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1 or os.environ.get("SMH_DIST_SINGLE_RANK") == "1":
            return dist
    return None


def _host_collective(values, op, dist, src=None):
    """A small host-side float64 vector through the process group (RCCL needs device tensors, gloo takes host ones):
    op = 'sum' | 'max' all-reduce, or 'bcast' from rank `src`.  Returns a numpy array identical on every rank."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(np.asarray(values, dtype=np.float64), dtype=torch.float64, device=dev)
    if op == "bcast":
        dist.broadcast(t, src=src or 0)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
    return t.cpu().numpy()


def train_head_bit(h):
    return 4 << h


class History(Callback):
    def __init__(self):
        super().__init__()
        self.history = {}
        self.epoch = []

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


def _as_callbacks(callbacks, csv_log, checkpoint_path, early_stopping):
    """`callbacks=[...]` as the reference passes them, plus the keyword spellings round 1 offered."""
    cbs = list(callbacks or [])
    for cb in cbs:
        if not isinstance(cb, Callback):
            raise TypeError("fit(callbacks=...): %r is not one of sm_hpss_mtl_amd.callbacks (EarlyStopping, ModelCheckpoint, "
                            "CSVLogger or a Callback subclass)" % (cb,))
    if csv_log:
        cbs.append(CSVLogger(csv_log))
    if early_stopping:
        cbs.append(EarlyStopping(**early_stopping))
    if checkpoint_path:
        mon = (early_stopping or {}).get("monitor", "val_loss")
        cbs.append(ModelCheckpoint(checkpoint_path, monitor=mon, save_best_only=True, save_weights_only=True))
    return cbs


class TrainingMixin:
    """Mixed into sm_hpss_mtl_amd.model.B3MTL (and, through CnnTrainingMixin, into CnnMTL)."""

    # C entry points of this model family's trainer: create, destroy, copy_state, grad_ptr, bucket_floats
    _TRAINER_API = ("smh_trainer_create", "smh_trainer_destroy", "smh_trainer_copy_state", "smh_trainer_grad_ptr",
                    "smh_trainer_bucket_floats")
    _MIN_TRAINER_CAP = 64

    # ---- optimiser state --------------------------------------------------------------------
    def _init_training_state(self):
        self._trainer = None
        self._trainer_cap = 0
        self._grad_view = None
        self.iterations = 0
        self.stop_training = False
        # lib/proposed_architectures.py:156-158
        self.optimizer = _opt.SGD(learning_rate=_opt.ExponentialDecay(self.initial_learning_rate, 3 * max(int(self.TR_STEPS), 1), 0.1),
                                  clipnorm=1, momentum=0.9)
        self._mask_seed = 1234 + int(os.environ.get("RANK", "0"))  # data parallel: every rank draws its own masks
        self._mask_calls = 0
        # bit-reproducible weight gradients (include/smh.h: smh_trainer_set_deterministic); SMH_DETERMINISTIC=1 turns it on for
        # every model of the process, `model.deterministic_gradients = True` for one
        self._deterministic = os.environ.get("SMH_DETERMINISTIC", "0") == "1"

    def learning_rate(self, step=None):
        """Learning rate of optimiser step `step` (default: the next one), e.g. ExponentialDecay(0.002, 3*TR_STEPS, 0.1)."""
        return self.optimizer.lr_at(self.iterations if step is None else step)

    # round-1 attribute names, still read by tools/ and tests
    @property
    def momentum(self):
        return getattr(self.optimizer, "momentum", 0.0)

    @property
    def clipnorm(self):
        return self.optimizer.clipnorm or 0.0

    def compile(self, loss=None, optimizer=None, metrics=None, loss_weights=None, **kwargs):
        """`model.compile(loss={...}, optimizer=optimizers.SGD(...), metrics={'3C': 'accuracy'})` as the reference calls it
        after reloading a model (Proposed_Work_Results.py:386-441).  The losses of the MTL graph are fixed (S/M/N binary
        cross-entropy, R mean squared error, 3C categorical cross-entropy): anything else is an error, not a silent
        change.  `optimizer` is one of sm_hpss_mtl_amd.optimizers."""
        if kwargs:
            raise TypeError("compile: unsupported arguments %s" % sorted(kwargs))
        if loss is not None:
            if isinstance(loss, str):
                loss = {n: loss for n in self.output_names} if len(self.output_names) == 1 else None
                if loss is None:
                    raise ValueError("compile: the MTL model needs one loss per output: %s" % _LOSS_OF)
            for name, fn in loss.items():
                if name not in self.output_names:
                    raise ValueError("compile: unknown output %r (outputs: %s)" % (name, self.output_names))
                if fn != _LOSS_OF[name]:
                    raise ValueError("compile: output %r is built with %s, not %r" % (name, _LOSS_OF[name], fn))
        if optimizer is not None:
            if not isinstance(optimizer, _opt._Optimizer):
                raise TypeError("compile: optimizer must be sm_hpss_mtl_amd.optimizers.SGD / Adam / Nadam, got %r" % (optimizer,))
            self._set_optimizer(optimizer)
        if loss_weights is not None:
            self.loss_weights = dict(loss_weights)
        if metrics is not None:
            m = metrics if isinstance(metrics, dict) else {"3C": metrics}
            for name, v in m.items():
                vv = v if isinstance(v, str) else (v[0] if len(v) == 1 else None)
                if name != "3C" or vv not in ("accuracy", "acc"):
                    raise ValueError("compile: the only metric of the MTL models is {'3C': 'accuracy'}, got %r" % (metrics,))

    def _set_optimizer(self, optimizer):
        self.optimizer = optimizer
        self.iterations = 0
        self._reset_optimizer_state()  # a freshly compiled Keras model starts a fresh optimiser

    def _reset_optimizer_state(self):
        if self._trainer is not None:
            _lib.check(self.lib.smh_trainer_reset_state(self._trainer, _cur_stream()), "smh_trainer_reset_state")

    def _head_spec(self):
        from .model import head_spec
        return head_spec(self.n_classes)

    def _trainer_fn(self, i):
        return getattr(self.lib, self._TRAINER_API[i])

    def _get_trainer(self, n):
        """The native trainer, grown when a batch exceeds its capacity.  Only the activation scratch depends on the
        capacity: momentum / Adam moments / step counters are copied into the larger trainer, so a bigger batch later
        in training never resets the optimiser."""
        if self._trainer is None or n > self._trainer_cap:
            cap = max(n, self._MIN_TRAINER_CAP)
            h = C.c_void_p()
            _lib.check(self._trainer_fn(0)(self._h, cap, C.byref(h)), self._TRAINER_API[0])
            if self._trainer is not None:
                _lib.check(self._trainer_fn(2)(h, self._trainer, _cur_stream()), self._TRAINER_API[2])
                self._trainer_fn(1)(self._trainer)
            self._trainer, self._trainer_cap = h, cap
            self._grad_view = None
            self._on_new_trainer()
        return self._trainer

    def _on_new_trainer(self):
        self._apply_deterministic()
        self._apply_train_dtype()

    def _apply_train_dtype(self):
        if self._trainer is not None and self._TRAINER_API[0] == "smh_trainer_create":
            _lib.check(self.lib.smh_trainer_set_dtype(self._trainer, 1 if getattr(self, "_train_dtype", "f32") == "bf16" else 0),
                       "smh_trainer_set_dtype")

    @property
    def train_dtype(self):
        """'f32' (default) or 'bf16': the training step's forward on the bf16 matrix pipe with split operands (f32-grade products,
        f32 accumulators and master weights; include/smh.h: smh_trainer_set_dtype).  B3_MTL only."""
        return getattr(self, "_train_dtype", "f32")

    @train_dtype.setter
    def train_dtype(self, dtype):
        if dtype not in ("f32", "bf16"):
            raise ValueError("train_dtype must be 'f32' or 'bf16', got %r" % (dtype,))
        if dtype == "bf16" and self._TRAINER_API[0] != "smh_trainer_create":
            raise ValueError("train_dtype='bf16' exists for the B3_MTL trainer only")
        self._train_dtype = dtype
        self._apply_train_dtype()

    def _apply_deterministic(self):
        if self._trainer is not None and hasattr(self.lib, "smh_trainer_set_deterministic") and self._TRAINER_API[0] == "smh_trainer_create":
            _lib.check(self.lib.smh_trainer_set_deterministic(self._trainer, 1 if self._deterministic else 0, _cur_stream()),
                       "smh_trainer_set_deterministic")

    @property
    def deterministic_gradients(self):
        """Weight gradients summed in 64-bit fixed point (integer atomics) instead of float atomics: bit-identical from run to
        run.  The Conv2D baselines' trainer sums in ordered partials and is deterministic as it stands."""
        return self._deterministic

    @deterministic_gradients.setter
    def deterministic_gradients(self, on):
        self._deterministic = bool(on)
        self._apply_deterministic()

    def _bucket_tensor(self):
        """torch view of the trainer's data-parallel bucket [flat gradient | BatchNorm batch statistics]."""
        if self._grad_view is None:
            ptr = self._trainer_fn(3)(self._trainer)
            n = int(self._trainer_fn(4)(self._trainer))

            class _Holder:
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
            self._grad_view = torch.as_tensor(_Holder(), device="cuda")
        return self._grad_view

    def _grad_tensor(self):
        """The flat gradient (canonical parameter order) of the last step."""
        return self._bucket_tensor()[: self.count_params()]

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

    def _loss_weight_array(self, only=None):
        w = [1.0] * len(self.output_names)
        if self.loss_weights:
            for i, n in enumerate(self.output_names):
                w[i] = float(self.loss_weights.get(n, 1.0))
        if only is not None:  # single-output sub-model: the other outputs are not part of it
            w = [1.0 if n == only else 0.0 for n in self.output_names]
        return (C.c_float * len(w))(*w)

    def _n_losses(self):
        return 3 * (len(self.output_names) - 1) + 4

    def losses_to_list(self, raw):
        """Raw device losses of one step (or their mean over steps) -> Keras order [loss, <per-output losses>, 3C_accuracy]."""
        lv = raw.detach().cpu().numpy() if isinstance(raw, torch.Tensor) else np.asarray(raw)
        nh = len(self.output_names) - 1
        return [float(lv[nh + 1] + lv[nh + 3])] + [float(v) for v in lv[: nh + 1]] + [float(lv[nh + 2])]

    # ---- one step ---------------------------------------------------------------------------------
    def train_on_batch(self, x, y, drop_tcn="auto", drop_heads="auto", apply=True, sync=True, _only=None, _mask=TRAIN_ALL):
        """One optimiser step.  Returns [loss, <per-output losses>, 3C_accuracy] like Keras; with sync=False the raw
        device tensor of the step's losses (no host round trip: feed it to `losses_to_list` later).
        drop_*: "auto" draws masks with the model's rates, None disables dropout, or pass mask tensors."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(device="cuda", dtype=torch.float32).contiguous()
        n = x.shape[0]
        if x.dim() != 3 or x.shape[1] != self.patch_size or x.shape[2] != self.n_feat:
            raise ValueError("expected input (N, %d, %d), got %s" % (self.patch_size, self.n_feat, tuple(x.shape)))
        if getattr(self, "block_variant", 0) != 0:
            raise NotImplementedError("training is built for the keras-tcn 2.3.x block (tcn_block='2.3'); the 2.8 block is inference only")
        yt = y if (isinstance(y, torch.Tensor) and y.is_cuda and y.dim() == 2) else self.pack_targets(y)
        if yt.shape[0] != n:
            raise ValueError("%d inputs but %d target rows" % (n, yt.shape[0]))
        self._sync_weights()
        tr = self._get_trainer(n)
        n_blocks, n_heads = self.nb_stacks * self.n_dilations, len(self.output_names) - 1
        if isinstance(drop_tcn, str) or isinstance(drop_heads, str):
            # both masks from ONE launch (csrc/smh_rng.hip; until round 3 a torch Bernoulli draw and a scaling): Philox keyed by this
            # replica's seed (1234 + RANK: every rank draws its own masks), one stream per step
            from .device_rng import dropout_masks
            n_t, n_h = n * n_blocks * 32, n * n_heads * 16
            masks = dropout_masks(n_t, 1.0 - self.dropout_rate, n_h, 1.0 - HEAD_DROPOUT, self._mask_seed, self._mask_calls)
            self._mask_calls += 1
            if isinstance(drop_tcn, str):
                drop_tcn = masks[:n_t].view(n, n_blocks, 32)
            if isinstance(drop_heads, str):
                drop_heads = masks[n_t:].view(n, n_heads, 16)
        losses = torch.empty(self._n_losses(), dtype=torch.float32, device="cuda")
        p = lambda t: None if t is None else C.c_void_p(t.contiguous().data_ptr())  # noqa: E731
        _lib.check(self.lib.smh_train_step_f32(tr, p(x), p(yt), n, p(drop_tcn), p(drop_heads), self._loss_weight_array(_only),
                                               p(losses), _cur_stream()), "smh_train_step_f32")
        if apply:
            self.apply_gradients(_mask)
        if not sync:
            return losses
        self._check_device_status()  # the step's forward may have given up on the device: never report its losses as a step
        return self.losses_to_list(losses)

    def _apply_native(self, lr, scale, mask):
        o = self.optimizer
        kind = {"sgd": 0, "adam": 1, "nadam": 2}[o.kind]
        b1 = o.momentum if kind == 0 else o.beta_1
        _lib.check(self.lib.smh_trainer_apply_f32(self._trainer, kind, lr, b1, getattr(o, "beta_2", 0.0),
                                                  getattr(o, "epsilon", 0.0), o.clipnorm or 0.0, scale, mask, _cur_stream()),
                   "smh_trainer_apply_f32")

    def apply_gradients(self, mask=TRAIN_ALL):
        """All-reduce the bucket [gradient | BatchNorm batch statistics] (if torch.distributed is initialised) -- SUM over
        ranks, the 1/world factor rides in the update --, then clip (after averaging: SURVEY 8e), update, repack."""
        scale = 1.0
        dist = process_group()
        if dist is not None:
            dist.all_reduce(self._bucket_tensor(), op=dist.ReduceOp.SUM)  # ONE flat bucket (0.9 MB) over RCCL
            scale = 1.0 / dist.get_world_size()
        self._apply_native(self.learning_rate(), scale, mask)
        self.iterations += 1
        self._device_newer = True

    def _l2_penalty(self):
        w = self.get_weights_dict()
        return float(sum(L2 * np.sum(w[n + "/dense/kernel"].astype(np.float64) ** 2) for n in self.output_names[:-1]))

    # ---- evaluate ---------------------------------------------------------------------------------
    def _losses_inference(self, x, y):
        """Losses / accuracy of one batch in inference mode (Keras `evaluate` semantics).  The heads' kernel-regulariser penalty
        depends on the weights only: `evaluate` computes it once for all its batches (`_eval_l2`) instead of downloading the weight
        vector behind every one of them."""
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
        cache = getattr(self, "_eval_l2", None)  # a dict while `evaluate` runs: the penalty is computed by its first batch
        if cache is None:
            l2 = self._l2_penalty()
        else:
            if "v" not in cache:
                cache["v"] = self._l2_penalty()
            l2 = cache["v"]
        return [sum(a * b for a, b in zip(lw, per)) + l2] + per + [acc]

    def evaluate(self, x=None, y=None, steps=None, verbose=0, batch_size=None, **kwargs):
        """model.evaluate(generator, steps) or evaluate(x, y) -> list matching `metrics_names`."""
        bad = [k for k in kwargs if k not in _FIT_IGNORED + ("callbacks", "return_dict")]
        if bad or kwargs.get("return_dict"):
            raise TypeError("evaluate: unsupported arguments %s" % sorted(bad or ["return_dict"]))
        # Row-weighted sums: under data parallel every rank evaluates its own shard of every batch and the all-reduce below
        # turns the shard means into the mean over the global batch -- the number a single device would report, and the SAME
        # number on every rank (EarlyStopping / ModelCheckpoint decide on it).  Single process: a plain mean over the batches.
        # The weights are rows / unit, unit = the largest first shard over the ranks (one MAX all-reduce per call): equal shards
        # weigh exactly 1.0, so the sums -- and with one rank the result, to the last bit -- are the single process's
        # ((v * rows) / rows is not v in floating point).
        dist = process_group()
        unit = []

        def weight(rows):
            if dist is None:
                return 1.0
            if not unit:
                unit.append(max(float(_host_collective([float(rows)], "max", dist)[0]), 1.0))
            return float(rows) / unit[0]

        if self._device_evaluate_ok():
            tot, cnt = self._evaluate_device(x, y, steps, weight)
            if dist is not None:
                red = _host_collective(np.concatenate([tot, [cnt]]), "sum", dist)
                tot, cnt = red[:-1], red[-1]
            return list(tot / max(cnt, 1.0))
        self._eval_l2 = {}  # the weights do not change while evaluating: one penalty for all batches (see _losses_inference)
        try:
            if y is not None:
                v = np.array(self._losses_inference(x, y), np.float64)
                rows = float(len(x))
                tot, cnt = v * rows, rows
            else:
                if steps is None:
                    raise ValueError("evaluate(generator) needs steps=")
                tot, cnt = None, 0.0
                for _ in range(int(steps)):
                    bx, by = next(x)
                    rows = weight(len(bx))  # Keras averages the batch values (equal batch sizes)
                    v = np.array(self._losses_inference(bx, by), np.float64) * rows
                    tot = v if tot is None else tot + v
                    cnt += rows
        finally:
            self._eval_l2 = None
        if dist is not None:
            red = _host_collective(np.concatenate([tot, [cnt]]), "sum", dist)
            tot, cnt = red[:-1], red[-1]
        return list(tot / max(cnt, 1.0))

    def _device_evaluate_ok(self):
        """B3_MTL with libsmh's evaluate kernel: the per-batch losses are summed on the device (smh_model_eval_losses_f32)."""
        return (getattr(self, "_TRAINER_API", ("",))[0] == "smh_trainer_create" and hasattr(self, "forward_device")
                and hasattr(getattr(self, "lib", None), "smh_model_eval_losses_f32") and getattr(self, "block_variant", 0) == 0
                and os.environ.get("SMH_EVAL_HOST", "0") != "1")

    def _evaluate_device(self, x, y, steps, weight):
        """evaluate() without a host round trip per batch: forward, then `smh_model_eval_losses_f32` adds the batch's mean losses and
        accuracy (float64, Keras' clipping: the arithmetic of `_losses_inference`) to device sums; ONE read-back at the end.
        Returns (unnormalised sums in metrics order, total weight) like the host loop.  SMH_EVAL_HOST=1 keeps the host loop."""
        nh = len(self.output_names) - 1
        sums = torch.zeros(nh + 3, dtype=torch.float64, device="cuda")  # [total | per-output losses | 3C accuracy]
        cnt = 0.0
        lw = (C.c_double * (nh + 1))(*[float((self.loss_weights or {}).get(n, 1.0)) for n in self.output_names])
        l2 = float(self._l2_penalty())  # the weights do not change while evaluating

        def one(bx, by, weight_of):
            if isinstance(bx, np.ndarray):
                bx = torch.from_numpy(np.ascontiguousarray(bx, dtype=np.float32))
            out = self.forward_device(bx.to(device="cuda", dtype=torch.float32))
            tgt = by if (isinstance(by, torch.Tensor) and by.is_cuda and by.dim() == 2) else self.pack_targets(by)
            if tgt.shape[0] != out.shape[0]:
                raise ValueError("%d inputs but %d target rows" % (out.shape[0], tgt.shape[0]))
            w = float(weight_of(out.shape[0]))  # Keras averages the batch values; data parallel: row-weighted (see evaluate)
            _lib.check(self.lib.smh_model_eval_losses_f32(self._h, C.c_void_p(out.data_ptr()), C.c_void_p(tgt.data_ptr()), out.shape[0],
                                                          C.c_double(w), lw, C.c_double(l2), C.c_void_p(sums.data_ptr()), _cur_stream()),
                       "smh_model_eval_losses_f32")
            return w

        if y is not None:
            cnt = one(x, y, float)  # arrays: one batch, weighted by its rows like the host path
        else:
            if steps is None:
                raise ValueError("evaluate(generator) needs steps=")
            for _ in range(int(steps)):
                cnt += one(*next(x), weight)
        self.check_status()  # one synchronisation for the whole pass; a device-side give-up raises here
        return sums.cpu().numpy(), cnt  # already in metrics order: [loss, <per-output losses>, 3C_accuracy]

    def _check_device_status(self):
        """Raise if a kernel of this model set the device error word (B3_MTL: smh_model_status); models without one: nothing."""
        if hasattr(self, "check_status"):
            self.check_status()

    # ---- fit ------------------------------------------------------------------------------------------
    def _train_step_raw(self, bx, by):
        return self.train_on_batch(bx, by, sync=False)

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, validation_data=None,
            steps_per_epoch=None, validation_steps=None, initial_epoch=0, csv_log=None, checkpoint_path=None,
            early_stopping=None, **kwargs):
        """`model.fit(generator, steps_per_epoch=, validation_data=generator, validation_steps=, epochs=, verbose=,
        callbacks=[csv_logger, es, mcp])` (Proposed_Work_Results.py:298-307), or arrays (x, y, batch_size).
        callbacks: sm_hpss_mtl_amd.callbacks.{EarlyStopping, ModelCheckpoint, CSVLogger} (or Callback subclasses); any
        other object, and any keyword this loop does not implement, raises instead of being dropped.
        csv_log= / checkpoint_path= / early_stopping=dict(...) are shorthand for the same three callbacks."""
        bad = [k for k in kwargs if k not in _FIT_IGNORED]
        if bad:
            raise TypeError("fit: unsupported arguments %s" % sorted(bad))
        hist = History()
        cbs = _as_callbacks(callbacks, csv_log, checkpoint_path, early_stopping) + [hist]
        names = self.metrics_names
        arrays = y is not None
        if arrays:
            xs = np.asarray(x, np.float32)
            yl = y if isinstance(y, (list, tuple)) else ([y] if isinstance(y, np.ndarray) else [y[k] for k in self.output_names])
            bs = batch_size or 32
            steps_per_epoch = steps_per_epoch or int(np.ceil(len(xs) / bs))
        elif steps_per_epoch is None:
            raise ValueError("fit(generator) needs steps_per_epoch=")
        self.stop_training = False
        dist = process_group()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
        for cb in cbs:
            cb.set_model(self)
            cb.set_writer(rank == 0)
            cb.on_train_begin()
        # Generator input on a GPU: the NEXT batch is drawn on a side stream as soon as this step's kernels are enqueued, so the front
        # end that builds it (and, data parallel, nothing else than it: the gradient all-reduce sits on the main stream) runs beside
        # the training step instead of behind it (SURVEY 8e "overlapped with the next micro-batch's HPSS").  The generator is
        # advanced exactly steps_per_epoch times per epoch, all of them before the validation pass, as without the side stream:
        # nothing is drawn across an epoch's end.  SMH_FIT_PREFETCH=0 keeps everything on one stream.
        side = None
        if not arrays and torch.cuda.is_available() and os.environ.get("SMH_FIT_PREFETCH", "1") != "0":
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())  # whatever the generator's buffers were last used by

        def fetch():
            if side is None:
                return next(x) + (None,)
            with torch.cuda.stream(side):
                bx_, by_ = next(x)
                ev = torch.cuda.Event()
                ev.record(side)
            return bx_, by_, ev

        def adopt(bx_, by_, ev):  # the batch was built on the side stream: order it before, and keep it alive for, the main stream
            if ev is None:
                return
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            vals = [bx_] + (list(by_.values()) if isinstance(by_, dict) else list(by_) if isinstance(by_, (list, tuple)) else [by_])
            for t in vals:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(main)

        for ep in range(int(initial_epoch), int(epochs)):
            t0 = time.time()
            acc = None  # device-side sum of the raw per-step losses: one read-back per epoch
            pending = None
            if side is not None:
                # whatever the main stream did with the generator's shared state since the last fetch (validation generator, a
                # callback running the front end on a file: the cached Frontend's workspace, the device featuregram cache) is
                # ordered before this epoch's first side-stream fetch
                side.wait_stream(torch.cuda.current_stream())
            for s in range(int(steps_per_epoch)):
                if arrays:
                    sl = slice((s * bs) % len(xs), (s * bs) % len(xs) + bs)
                    bx, by = xs[sl], [np.asarray(a)[sl] for a in yl]
                    if world > 1:  # data parallel: this rank's rows of the global batch (round-robin, like the generator)
                        from .sharding import shard_indices
                        mine = shard_indices(len(bx), rank, world)
                        bx, by = bx[mine], [a[mine] for a in by]
                else:
                    bx, by, ev = pending if pending is not None else fetch()
                    pending = None
                    adopt(bx, by, ev)
                raw = self._train_step_raw(bx, by)
                if side is not None and s + 1 < int(steps_per_epoch):
                    pending = fetch()
                acc = raw.clone() if acc is None else acc.add_(raw)
            mean_raw = acc / float(max(int(steps_per_epoch), 1))
            if dist is not None:  # the training loss of the GLOBAL batch: mean over the ranks' (equal-sized) shards
                dist.all_reduce(mean_raw, op=dist.ReduceOp.SUM)
                mean_raw = mean_raw / float(world)
            mean = self.losses_to_list(mean_raw)
            self._check_device_status()  # once per epoch, where the losses are read back anyway (the steps only enqueued work)
            logs = {n: float(v) for n, v in zip(names, mean)}
            if validation_data is not None:
                if isinstance(validation_data, (tuple, list)) and not hasattr(validation_data, "__next__"):
                    val = self.evaluate(validation_data[0], validation_data[1])
                else:
                    val = self.evaluate(validation_data, steps=validation_steps)
                logs.update({"val_" + n: float(v) for n, v in zip(names, val)})
            if dist is not None:
                # the all-reduces above give every rank the same values already; the broadcast makes "identical logs" a
                # property of the protocol rather than of the backend's reduction order: rank 0's numbers are THE logs
                keys = list(logs)
                vals = _host_collective([logs[k] for k in keys], "bcast", dist, src=0)
                logs = {k: float(v) for k, v in zip(keys, vals)}
            if verbose and rank == 0:
                print("Epoch %d/%d - %.1fs - " % (ep + 1, epochs, time.time() - t0) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
            for cb in cbs:
                cb.on_epoch_end(ep, logs)
            if dist is not None:  # a stop decided on ANY rank (a user callback with rank-local state) stops every rank
                self.stop_training = bool(_host_collective([1.0 if self.stop_training else 0.0], "max", dist)[0] > 0.5)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end()
        return hist
