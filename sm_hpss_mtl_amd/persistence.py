"""Model persistence either side of the path, as the reference drives it
(Proposed_Work_Results.py:370-384):

    model.save_weights(weightFile); open(architechtureFile, 'w').write(model.to_json())
    np.savez(paramFile, epochs=..., batch_size=..., lr=..., trainingTimeTaken=...)
    ...
    model = model_from_json(open(architechtureFile).read()); model.load_weights(weightFile)

`save_weights(path)`: a path ending in `.h5` / `.hdf5` is written as a real HDF5 file in Keras' weight-file layout
through the system's libhdf5 (`h5io.py`; h5py itself is absent here); anything else -- or a machine without libhdf5 --
becomes `<path>.npz` (tensor names with '/' spelled '__', canonical order).  `load_weights(path)` takes either.
`to_json()` carries the constructor arguments, and `model_from_json` below rebuilds the object.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import h5io


def save_weights_file(path, weights):
    """weights: ordered mapping canonical tensor name -> array.  Returns the path written."""
    path = str(path)
    if path.endswith((".h5", ".hdf5")) and h5io.available():
        h5io.write_weights(path, weights)
        return path
    p = path if path.endswith(".npz") else path + ".npz"
    np.savez(p, **{k.replace("/", "__"): v for k, v in weights.items()})
    return p


def load_weights_file(path):
    """-> dict canonical tensor name -> array, from an HDF5 weight file or the .npz form."""
    path = str(path)
    if os.path.isfile(path):
        with open(path, "rb") as f:
            magic = f.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            layers, _ = h5io.read_weights(path)
            return {name: arr for ws in layers.values() for name, arr in ws.items()}
    p = path if path.endswith(".npz") else path + ".npz"
    with np.load(p) as z:
        return {k.replace("__", "/"): z[k] for k in z.files}


def model_from_json(text, seed=None):
    """Inverse of B3MTL.to_json / CnnMTL.to_json (tensorflow.keras.models.model_from_json at the call site)."""
    d = json.loads(text)
    name, cfg = d.get("class_name"), d.get("config", {})
    if name == "B3_MTL":
        from .model import B3MTL
        m = B3MTL(n_feat=cfg["n_feat"], patch_size=cfg["patch_size"], n_classes=cfg["n_classes"], seed=seed,
                  **{k: cfg[k] for k in ("nb_filters", "kernel_size", "nb_stacks", "n_dilations") if k in cfg})
        if "dropout_rate" in cfg:  # drawn at build time by the reference (proposed_architectures.py:136): part of the architecture
            m.dropout_rate = float(cfg["dropout_rate"])
        return m
    if name == "B3_MTL_head":  # the single-output sub-model the DAFx driver saves (DAFx12...:551-552, 566-567)
        head = cfg["head"]
        parent = model_from_json(json.dumps({"class_name": "B3_MTL", "config": {k: v for k, v in cfg.items() if k != "head"}}), seed=seed)
        return HeadModel(parent, head)
    if name in ("Doukhan_MTL", "Papakostas_MTL", "Jang_MTL"):
        from .cnn_models import CnnMTL
        return CnnMTL(name[:-4], tuple(cfg["input_shape"]), n_classes=cfg["n_classes"], seed=seed, n_mels=cfg.get("n_mels", 120),
                      n_fft=cfg.get("n_fft", 512), fs=cfg.get("fs", 16000), fc_width=cfg.get("fc_width", 0))
    raise ValueError("model_from_json: unknown class_name %r" % (name,))


class _LayerRef:
    """`model.get_layer(name)`: only the named outputs of the MTL graphs can be addressed ('S', 'M', 'N', 'R', '3C')."""

    def __init__(self, model, name):
        if name not in model.output_names:
            raise ValueError("get_layer(%r): the layers that can be addressed are the outputs %s" % (name, model.output_names))
        self.model, self.name = model, name
        self.output = self  # what Model(inputs, outputs) receives


class HeadModel:
    """`Model(trained_model.input, trained_model.get_layer('M').output)` of the DAFx driver
    (DAFx12_Speech_Music_Detection_B3_MTL_v2.py:518-523): the same network, ONE output -- the TCN trunk plus the Dense(16) /
    BatchNorm / Dense(1, sigmoid) of that head.  It shares the weights (and the native trainer) of the model it was cut
    from, exactly like the Keras sub-model shares layers.

    Inference: `predict`.  Fine-tuning as the driver does it (:524-571): `compile(loss='binary_crossentropy',
    optimizer=optimizers.Nadam(learning_rate=0.002), metrics='accuracy')`, then `fit(generator, steps_per_epoch=, ...,
    callbacks=[...])` / `train_on_batch` / `evaluate`; metrics_names = ['loss', 'accuracy'].  Only the tensors of the
    sub-model are updated (smh_trainer_apply_f32 with active_mask = trunk | this head); the loss is the head's binary
    cross-entropy plus the l2(0.01) penalty of its own Dense(16) kernel."""

    def __init__(self, model, name):
        from . import optimizers as _opt
        self.model, self.name = model, name
        self._index = model.output_names.index(name)
        self.output_names = [name]
        self.input = model.input
        self.optimizer = None
        self.iterations = 0
        self.stop_training = False
        self._opt = _opt

    # ---- inference ----
    def predict(self, x, batch_size=None, verbose=0, **kw):
        return self.model.predict(x, **kw)[self._index]

    def summary(self, print_fn=print):
        print_fn("Model: output %r of" % self.name)
        self.model.summary(print_fn=print_fn)

    # ---- training surface ----
    @property
    def metrics_names(self):
        return ["loss", "accuracy"]

    def compile(self, loss=None, optimizer=None, metrics=None, **kwargs):
        if kwargs:
            raise TypeError("compile: unsupported arguments %s" % sorted(kwargs))
        if self.name not in ("S", "M", "N"):
            raise ValueError("only the sigmoid heads S / M / N can be trained as a sub-model (the driver cuts out 'M' or 'S')")
        if loss not in (None, "binary_crossentropy"):
            raise ValueError("compile: head %r is a sigmoid output: loss must be 'binary_crossentropy', got %r" % (self.name, loss))
        mm = [metrics] if isinstance(metrics, str) else list(metrics or [])
        if any(v not in ("accuracy", "acc") for v in mm):
            raise ValueError("compile: the sub-model reports 'accuracy' (binary, threshold 0.5) only, got %r" % (metrics,))
        if optimizer is not None:
            if not isinstance(optimizer, self._opt._Optimizer):
                raise TypeError("compile: optimizer must be one of sm_hpss_mtl_amd.optimizers")
            self.optimizer = optimizer
            self.iterations = 0
            self.model._reset_optimizer_state()  # a freshly compiled Keras model starts a fresh optimiser

    def _mask(self):
        from .training import TRAIN_TRUNK, train_head_bit
        return TRAIN_TRUNK | train_head_bit(self._index)

    def _targets(self, y):
        import numpy as np
        import torch
        if isinstance(y, dict):
            y = y[self.name]
        if isinstance(y, (list, tuple)) and len(y) == 1:
            y = y[0]
        yt = torch.zeros((len(y), self.model.out_dim), dtype=torch.float32)
        col = sum(od for _, od, _ in self.model._head_spec()[: self._index])
        yt[:, col] = torch.as_tensor(np.asarray(y, dtype=np.float32).reshape(-1))
        return yt.cuda()

    def train_on_batch(self, x, y, sync=True, **drop):
        """One Nadam (or whatever was compiled) step on the sub-model.  Returns [loss, accuracy]."""
        if self.optimizer is None:
            raise RuntimeError("compile(optimizer=...) the sub-model before training it")
        m = self.model
        saved, saved_it = m.optimizer, m.iterations
        m.optimizer, m.iterations = self.optimizer, self.iterations
        try:
            raw = m.train_on_batch(x, self._targets(y), sync=False, _only=self.name, _mask=self._mask(), **drop)
        finally:
            m.optimizer, m.iterations = saved, saved_it
        self.iterations += 1
        return self.losses_to_list(raw) if sync else raw

    def _train_step_raw(self, bx, by):
        return self.train_on_batch(bx, by, sync=False)

    def fit(self, *args, **kwargs):
        """fit(generator | arrays, ..., callbacks=[...]): the loop of TrainingMixin.fit on the single-output sub-model."""
        from .training import TrainingMixin
        return TrainingMixin.fit(self, *args, **kwargs)

    def losses_to_list(self, raw):
        import numpy as np
        lv = raw.detach().cpu().numpy() if hasattr(raw, "detach") else np.asarray(raw)
        nh, h = len(self.model.output_names) - 1, self._index
        return [float(lv[h] + lv[nh + 4 + h]), float(lv[2 * nh + 4 + h])]

    def evaluate(self, x=None, y=None, steps=None, verbose=0, batch_size=None, **kwargs):
        """[loss, accuracy] in inference mode: evaluate(x, y) or evaluate(generator, steps=)."""
        import numpy as np
        if kwargs:
            raise TypeError("evaluate: unsupported arguments %s" % sorted(kwargs))

        def one(bx, by):
            o = self.predict(bx).astype(np.float64).reshape(-1)
            t = np.asarray(by, np.float64).reshape(-1)
            eps = 1e-7
            oc = np.clip(o, eps, 1 - eps)
            bce = float(np.mean(-(t * np.log(oc + eps) + (1 - t) * np.log(1 - oc + eps))))
            w = self.model.get_weights_dict()[self.name + "/dense/kernel"].astype(np.float64)
            return np.array([bce + 0.01 * float(np.sum(w * w)), float(np.mean((o > 0.5) == (t > 0.5)))])
        if y is not None:
            return list(one(x, y))
        if steps is None:
            raise ValueError("evaluate(generator) needs steps=")
        tot = sum(one(*next(x)) for _ in range(int(steps)))
        return list(tot / max(int(steps), 1))

    # ---- weights: the sub-model's own tensors (trunk + this head), as a Keras sub-model would save them ----
    def _own(self, name):
        return name.startswith("tcn/") or name.startswith(self.name + "/")

    def get_weights(self):
        return self.model.get_weights()

    def set_weights(self, arrays):
        self.model.set_weights(arrays)

    def save_weights(self, path):
        w = self.model.get_weights_dict()
        return save_weights_file(path, {k: v for k, v in w.items() if self._own(k)})

    def load_weights(self, path):
        w = dict(self.model.get_weights_dict())
        got = load_weights_file(path)
        unknown = [k for k in got if k not in w]
        if unknown:
            raise ValueError("load_weights: tensors %s do not belong to this model" % unknown[:4])
        w.update(got)
        self.model.set_weights_dict(w)

    def to_json(self):
        d = json.loads(self.model.to_json())
        return json.dumps({"class_name": "B3_MTL_head", "config": dict(d["config"], head=self.name)})


def Model(inputs, outputs):
    """tensorflow.keras.models.Model at the one call shape the reference uses on a trained MTL model."""
    if isinstance(outputs, _LayerRef) and inputs is outputs.model.input:
        return HeadModel(outputs.model, outputs.name)
    raise TypeError("Model(inputs, outputs): expected (model.input, model.get_layer(name).output) of one of our models")


class ModelSurfaceMixin:
    """`input` and `get_layer` for B3MTL / CnnMTL."""

    @property
    def input(self):
        return self  # identity token: only ever passed back to Model()

    def get_layer(self, name):
        return _LayerRef(self, name)
