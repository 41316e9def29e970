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
from collections import OrderedDict

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


def load_weights_file(path, arch_json=None):
    """-> dict canonical tensor name -> array, from an HDF5 weight file or the .npz form.  An HDF5 file written by Keras
    itself (auto-generated layer names) is mapped onto the canonical names (`keras_layers_to_canonical`), with the help of
    the architecture JSON the reference saves next to it (`<name>.json`, Proposed_Work_Results.py:372-373) when found."""
    path = str(path)
    if os.path.isfile(path):
        with open(path, "rb") as f:
            magic = f.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            layers, _ = h5io.read_weights(path)
            flat = {name: arr for ws in layers.values() for name, arr in ws.items()}
            if any(k.startswith("tcn/") or "/dense/kernel" in k for k in flat) or any(k.startswith("conv1/") for k in flat):
                return flat  # written by save_weights of this package
            if arch_json is None and os.path.exists(os.path.splitext(path)[0] + ".json"):
                arch_json = os.path.splitext(path)[0] + ".json"
            arch = None
            if arch_json is not None:
                arch = json.loads(arch_json) if str(arch_json).lstrip().startswith("{") else json.load(open(arch_json))
            return keras_layers_to_canonical(layers, arch)
    p = path if path.endswith(".npz") else path + ".npz"
    with np.load(p) as z:
        return {k.replace("__", "/"): z[k] for k in z.files}


def keras_layers_to_canonical(layers, arch=None):
    """Weights of a B3_MTL model as Keras wrote them -> canonical tensor names of this package.

    layers: ordered mapping Keras layer name -> ordered mapping weight name -> array (`h5io.read_weights`), e.g.
    'tcn_initial_conv', 'tcn_dilated_conv_1_tanh_s0', 'conv1d_7', 'dense_3', 'batch_normalization_2', 'S', 'M', 'R', '3C'
    [recollection of keras-tcn 2.3.x / tf.keras layer naming; the exact names do not matter here].
    arch: the parsed `model.to_json()` of the Keras model, or None.

    The TCN's Conv1D layers are taken in file order (= creation order = graph order): initial 1x1 conv, then per residual
    block the dilated conv followed by its 1x1 conv; shapes are checked.  The output layers carry the names the reference
    gave them ('S', 'M', 'N', 'R', '3C': lib/proposed_architectures.py:52,65,78,150).  Which Dense(16) / BatchNormalization
    pair feeds which head is read off the architecture JSON (inbound_nodes: out Dense <- Dropout <- Activation <-
    BatchNormalization <- Dense(16) <- Flatten).  Without the JSON the heads' hidden layers are assigned in creation order
    (S, M, [N,] R -- the order MTL_modifications builds them in); the dead first 'x_mu' / 'x_smr' blocks (:55-58, :68-71)
    are not part of the saved graph."""
    convs, dense16, bns, named = [], [], [], {}
    for lname, ws in layers.items():
        arrs = list(ws.values())
        if not arrs:
            continue
        if lname in ("S", "M", "N", "R", "3C"):
            named[lname] = arrs
        elif len(arrs) == 2 and arrs[0].ndim == 3:
            convs.append((lname, arrs))
        elif len(arrs) == 2 and arrs[0].ndim == 2 and arrs[0].shape[1] == 16:
            dense16.append((lname, arrs))
        elif len(arrs) == 4 and all(a.shape == (16,) for a in arrs):
            bns.append((lname, arrs))
        else:
            raise ValueError("keras weight file: layer %r with shapes %s does not belong to a B3_MTL graph" % (lname, [a.shape for a in arrs]))
    if "3C" not in named or len(convs) < 3 or (len(convs) - 1) % 2:
        raise ValueError("keras weight file: expected an initial Conv1D, (dilated conv, 1x1 conv) pairs and a '3C' output layer")
    out = OrderedDict()
    out["tcn/initial_conv/kernel"], out["tcn/initial_conv/bias"] = convs[0][1]
    C = convs[0][1][0].shape[2]
    n_blocks = (len(convs) - 1) // 2
    n_dil = 8
    if n_blocks % n_dil:
        raise ValueError("keras weight file: %d residual blocks is not a multiple of the 8 dilations" % n_blocks)
    for b in range(n_blocks):
        (_, (k1, b1)), (_, (k2, b2)) = convs[1 + 2 * b], convs[2 + 2 * b]
        if k1.shape[1:] != (C, C) or k2.shape != (1, C, C):
            raise ValueError("keras weight file: block %d has kernels %s / %s" % (b, k1.shape, k2.shape))
        p = "tcn/s%d_d%d" % (b // n_dil, 2 ** (b % n_dil))
        out[p + "/conv/kernel"], out[p + "/conv/bias"], out[p + "/conv1x1/kernel"], out[p + "/conv1x1/bias"] = k1, b1, k2, b2
    out["3C/kernel"], out["3C/bias"] = named["3C"]
    heads = [h for h in ("S", "M", "N", "R") if h in named]
    if len(dense16) != len(heads) or len(bns) != len(heads):
        raise ValueError("keras weight file: %d heads but %d Dense(16) / %d BatchNormalization layers" % (len(heads), len(dense16), len(bns)))
    feeder = {}
    if arch is not None:
        cfg_layers = arch["config"]["layers"]
        inbound = {}
        for L in cfg_layers:
            src = []
            for node in L.get("inbound_nodes", []):
                for ref in (node if isinstance(node, list) else []):
                    if isinstance(ref, list) and ref and isinstance(ref[0], str):
                        src.append(ref[0])
            inbound[L["name"]] = src
        cls = {L["name"]: L["class_name"] for L in cfg_layers}
        for h in heads:
            cur, found = h, {}
            for _ in range(8):  # walk up: Dense(out) <- Dropout <- Activation <- BatchNormalization <- Dense(16)
                ups = inbound.get(cur, [])
                if len(ups) != 1:
                    break
                cur = ups[0]
                if cls.get(cur) == "BatchNormalization":
                    found["bn"] = cur
                elif cls.get(cur) == "Dense":
                    found["dense"] = cur
                    break
            if "bn" not in found or "dense" not in found:
                raise ValueError("architecture JSON: cannot trace head %r back to its BatchNormalization / Dense(16)" % h)
            feeder[h] = (found["dense"], found["bn"])
    else:
        for i, h in enumerate(heads):  # creation order
            feeder[h] = (dense16[i][0], bns[i][0])
    d16, bnd = dict(dense16), dict(bns)
    for h in heads:
        dn, bn = feeder[h]
        if dn not in d16 or bn not in bnd:
            raise ValueError("architecture JSON names layers %r / %r that the weight file does not hold" % (dn, bn))
        out[h + "/dense/kernel"], out[h + "/dense/bias"] = d16[dn]
        out[h + "/bn/gamma"], out[h + "/bn/beta"], out[h + "/bn/moving_mean"], out[h + "/bn/moving_variance"] = bnd[bn]
        out[h + "/out/kernel"], out[h + "/out/bias"] = named[h]
    return out


def model_from_json(text, seed=None):
    """Inverse of B3MTL.to_json / CnnMTL.to_json (tensorflow.keras.models.model_from_json at the call site)."""
    d = json.loads(text)
    name, cfg = d.get("class_name"), d.get("config", {})
    if name in ("Functional", "Model") and isinstance(cfg.get("layers"), list):
        # an architecture file Keras wrote for the reference's B3_MTL graph (Proposed_Work_Results.py:372-373): read the
        # constructor arguments off it -- Input (None, patch_size, N_MELS), units of '3C', the SpatialDropout1D rate
        L = {l["name"]: l for l in cfg["layers"]}
        inp = next(l for l in cfg["layers"] if l["class_name"] == "InputLayer")
        shp = inp["config"].get("batch_input_shape") or inp["config"].get("batch_shape")
        if "3C" not in L or len(shp) != 3:
            raise ValueError("model_from_json: this Keras architecture is not the B3_MTL graph (TCN input, '3C' output)")
        n_conv = sum(1 for l in cfg["layers"] if l["class_name"] == "Conv1D")
        if (n_conv - 1) % 16:
            raise ValueError("model_from_json: %d Conv1D layers: not 1 + 2 x (stacks x 8 dilations)" % n_conv)
        from .model import B3MTL
        m = B3MTL(n_feat=int(shp[2]), patch_size=int(shp[1]), n_classes=int(L["3C"]["config"]["units"]), seed=seed,
                  nb_stacks=(n_conv - 1) // 16)
        rates = [l["config"]["rate"] for l in cfg["layers"] if l["class_name"] == "SpatialDropout1D"]
        if rates:
            m.dropout_rate = float(rates[0])
        return m
    if name == "B3_MTL":
        from .model import B3MTL
        m = B3MTL(n_feat=cfg["n_feat"], patch_size=cfg["patch_size"], n_classes=cfg["n_classes"], seed=seed,
                  **{k: cfg[k] for k in ("nb_filters", "kernel_size", "nb_stacks", "n_dilations", "tcn_block") if k in cfg})
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

    def _check_device_status(self):
        self.model.check_status()  # the parent model's device error word (the sub-model runs the parent's kernels)

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
        # data parallel: row-weighted sums all-reduced, as in TrainingMixin.evaluate (the loss of the global batch on every rank)
        from .training import _host_collective, process_group
        dist = process_group()
        if y is not None:
            tot, cnt = one(x, y) * float(len(x)), float(len(x))
        else:
            if steps is None:
                raise ValueError("evaluate(generator) needs steps=")
            tot, cnt = np.zeros(2), 0.0
            for _ in range(int(steps)):
                bx, by = next(x)
                rows = float(len(bx)) if dist is not None else 1.0
                tot, cnt = tot + one(bx, by) * rows, cnt + rows
        if dist is not None:
            red = _host_collective(np.concatenate([tot, [cnt]]), "sum", dist)
            tot, cnt = red[:-1], red[-1]
        return list(tot / max(cnt, 1.0))

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
