"""Counterpart of lib/proposed_architectures.py for the hot path: `get_Lemaire_MTL_model` (B3_MTL).

Same signature and return value as the reference (proposed_architectures.py:85-91,170): a model object
with the Keras-style surface the drivers use, and the initial learning rate 0.002.  The 5-class variant
of 5_class_classification.py:220-308 is selected by n_classes=5.  The Conv2D MTL baselines (Doukhan /
Papakostas / Jang, row a13) are inference models (`sm_hpss_mtl_amd.cnn_models.CnnMTL`).
"""
from __future__ import annotations

from ..cnn_models import CnnMTL
from ..model import B3MTL


def get_Lemaire_MTL_model(TR_STEPS, N_MELS=120, n_classes=3, patch_size=68, loss_weights=None, seed=None, tcn_block="2.3"):
    """tcn_block: which residual block the third-party `tcn.TCN` (unpinned in the reference) builds -- "2.3" (default: the API
    the reference's positional call at :144 binds under) or "2.8" (two-convolution block, inference only)."""
    model = B3MTL(n_feat=N_MELS, patch_size=patch_size, n_classes=n_classes, TR_STEPS=TR_STEPS,
                  loss_weights=loss_weights, seed=seed, tcn_block=tcn_block)
    return model, model.initial_learning_rate


def get_Doukhan_MTL_model(PARAMS, n_classes=3, seed=None):
    """proposed_architectures.py:425-511 -> (model, 0.0001); input PARAMS['input_shape'][PARAMS['Model']] = (2F, W, 1)."""
    model = CnnMTL("Doukhan", PARAMS["input_shape"][PARAMS["Model"]], n_classes=n_classes, seed=seed)
    return model, model.initial_learning_rate


def get_Papakostas_MTL_model(PARAMS, n_classes=3, seed=None):
    """proposed_architectures.py:516-588 -> (model, 0.001)."""
    model = CnnMTL("Papakostas", PARAMS["input_shape"][PARAMS["Model"]], n_classes=n_classes, seed=seed)
    return model, model.initial_learning_rate


def get_Jang_MTL_model(PARAMS, fs=16000, Tw=25, n_mels=120, t_dim=5, n_classes=3, seed=None):
    """proposed_architectures.py:650-764 -> (model, 0.001); the mel-scale kernels start from the Slaney mel
    weights of librosa.filters.mel(fs, n_fft=PARAMS['n_fft'][Model], n_mels) like the reference's Constant
    initialiser."""
    if t_dim != 5:
        raise ValueError("the mel-scale layer is built for t_dim=5 (the reference's only value)")
    model = CnnMTL("Jang", PARAMS["input_shape"][PARAMS["Model"]], n_classes=n_classes, seed=seed, n_mels=n_mels,
                   n_fft=PARAMS["n_fft"][PARAMS["Model"]], fs=fs)
    return model, model.initial_learning_rate


# tensorflow.keras.models.model_from_json at the reference's call site (Proposed_Work_Results.py:381-383)
from ..persistence import Model, model_from_json  # noqa: E402,F401
