"""Counterpart of lib/proposed_architectures.py for the hot path: `get_Lemaire_MTL_model` (B3_MTL).

Same signature and return value as the reference (proposed_architectures.py:85-91,170): a model object
with the Keras-style surface the drivers use, and the initial learning rate 0.002.  The 5-class variant
of 5_class_classification.py:220-308 is selected by n_classes=5.  Doukhan / Papakostas / Jang MTL models
(Conv2D) are second-priority rows and not built yet.
"""
from __future__ import annotations

from ..model import B3MTL


def get_Lemaire_MTL_model(TR_STEPS, N_MELS=120, n_classes=3, patch_size=68, loss_weights=None, seed=None):
    model = B3MTL(n_feat=N_MELS, patch_size=patch_size, n_classes=n_classes, TR_STEPS=TR_STEPS,
                  loss_weights=loss_weights, seed=seed)
    return model, model.initial_learning_rate


def _not_built(name):
    def f(*a, **k):
        raise NotImplementedError("%s is a second-priority row of SURVEY 8(a13); only B3_MTL is built" % name)
    return f


get_Doukhan_MTL_model = _not_built("get_Doukhan_MTL_model")
get_Papakostas_MTL_model = _not_built("get_Papakostas_MTL_model")
get_Jang_MTL_model = _not_built("get_Jang_MTL_model")
