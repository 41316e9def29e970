"""Model persistence either side of the path, as the reference drives it
(Proposed_Work_Results.py:370-384):

    model.save_weights(weightFile); open(architechtureFile, 'w').write(model.to_json())
    np.savez(paramFile, epochs=..., batch_size=..., lr=..., trainingTimeTaken=...)
    ...
    model = model_from_json(open(architechtureFile).read()); model.load_weights(weightFile)

h5py is absent in this environment, so `save_weights(path)` writes `<path>.npz` (tensor names = the Keras weight names
with '/' spelled '__', canonical order) whatever extension the caller passes -- the reference's `.h5` path argument
works unchanged; `to_json()` carries the constructor arguments, and `model_from_json` below rebuilds the object.
"""
from __future__ import annotations

import json


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
