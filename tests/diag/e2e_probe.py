"""diagnostic: is the train/val gap of tests/test_driver_sequence_gpu.py overfitting or a train/inference mismatch?"""
import copy, pathlib, sys, tempfile
import numpy as np
sys.path.insert(0, ".")
from tests.test_driver_sequence_gpu import _folds, _params, _train_model
from lib.proposed_architectures import get_Lemaire_MTL_model
from sm_hpss_mtl_amd.generators import generator
tmp = pathlib.Path(tempfile.mkdtemp())
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 20
folder, train, test = _folds(tmp, n_files=nf)
P = _params(tmp, folder, copy.deepcopy(train), test)
np.random.seed(0)
model, _ = get_Lemaire_MTL_model(TR_STEPS=P['TR_STEPS'], N_MELS=240, n_classes=3, patch_size=68, seed=1)
if len(sys.argv) > 2:
    model.dropout_rate = float(sys.argv[2])
print("spatial dropout rate", model.dropout_rate)
model, H = _train_model(P, model, str(tmp / "w.h5"), str(tmp / "l.csv"))
h = H.history
print("train 3C_acc", np.round(h["3C_accuracy"], 3)); print("val   3C_acc", np.round(h["val_3C_accuracy"], 3))
print("val_loss", np.round(h["val_loss"], 3), "val_3C_loss", np.round(h["val_3C_loss"], 3), "val_R_loss", np.round(h["val_R_loss"], 3))
P2 = dict(P); P2["data_augmentation_with_noise"] = False
tr_files = {k: v for k, v in train.items()}
print("inference-mode evaluate on TRAIN files:", np.round(model.evaluate(generator(P2, folder, copy.deepcopy(tr_files), 16), steps=6), 3))
print("inference-mode evaluate on TEST  files:", np.round(model.evaluate(generator(P2, folder, copy.deepcopy(test), 16), steps=6), 3))
print(model.metrics_names)
