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
    (DAFx12_Speech_Music_Detection_B3_MTL_v2.py:518-523): the same network, one output.  Inference only (the
    driver's optional Nadam fine-tuning of the sub-model is not built)."""

    def __init__(self, model, name):
        self.model, self.name = model, name
        self._index = model.output_names.index(name)
        self.output_names = [name]
        self.input = model.input

    def predict(self, x, batch_size=None, verbose=0, **kw):
        return self.model.predict(x, **kw)[self._index]

    def summary(self, print_fn=print):
        print_fn("Model: output %r of" % self.name)
        self.model.summary(print_fn=print_fn)


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
