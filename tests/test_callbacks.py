"""Host logic of `model.fit(..., callbacks=[csv_logger, es, mcp])` -- the literal call of the reference's train_model
(Proposed_Work_Results.py:275-307) -- on a scripted stand-in model (no GPU): the loop, the three Keras callbacks with the
constructor arguments the reference uses, their tf.keras 2.x semantics, and loud failures for anything not implemented."""
import csv

import numpy as np
import pytest
import torch

from sm_hpss_mtl_amd.callbacks import Callback, CSVLogger, EarlyStopping, ModelCheckpoint
from sm_hpss_mtl_amd.training import TrainingMixin


class Scripted(TrainingMixin):
    """fit() of the product with the device step replaced by a script: val_loss of epoch e = script[e]."""
    metrics_names = ["loss", "S_loss", "M_loss", "R_loss", "3C_loss", "3C_accuracy"]
    output_names = ["S", "M", "R", "3C"]

    def __init__(self, script):
        self.script, self.epoch_seen, self.w = list(script), 0, np.zeros(3)
        self.saved = []
        self.stop_training = False
        self.steps = 0

    def _train_step_raw(self, bx, by):
        self.steps += 1
        self.w = self.w + 1.0  # "training" moves the weights every step
        return torch.tensor([0.1, 0.2, 0.3, 0.4, 1.0, 0.5, 0.05, 0, 0, 0, 0, 0, 0], dtype=torch.float32)

    def evaluate(self, x=None, y=None, steps=None, **kw):
        v = self.script[min(self.epoch_seen, len(self.script) - 1)]
        self.epoch_seen += 1
        return [v, 0.1, 0.2, 0.3, 0.4, 0.5]

    def get_weights(self):
        return [self.w.copy()]

    def set_weights(self, ws):
        self.w = ws[0].copy()

    def save_weights(self, path):
        self.saved.append((path, self.w.copy()))

    def to_json(self):
        return "{}"


def _gen():
    while True:
        yield np.zeros((2, 68, 240), np.float32), {}


def test_reference_train_model_call_sequence(tmp_path):
    weightFile, logFile = str(tmp_path / "model.h5"), str(tmp_path / "model_log.csv")
    # the three constructions of Proposed_Work_Results.py:276-278, verbatim arguments
    es = EarlyStopping(monitor='val_loss', mode='auto', verbose=1, restore_best_weights=True, min_delta=0.01, patience=5)
    mcp = ModelCheckpoint(weightFile, monitor='val_loss', verbose=0, save_best_only=True, save_weights_only=True, mode='auto', save_freq='epoch')
    csv_logger = CSVLogger(logFile)
    #          e0    e1     e2     e3     e4    e5    e6    e7    e8
    script = [1.00, 0.995, 0.96, 0.955, 0.97, 0.98, 0.99, 0.99, 0.99, 0.5, 0.4]
    model = Scripted(script)
    History = model.fit(_gen(), steps_per_epoch=3, validation_data=_gen(), validation_steps=2, epochs=50, verbose=1,
                        callbacks=[csv_logger, es, mcp])
    # EarlyStopping: improvements of more than 0.01 only at e0 (from inf) and e2 (0.96 < 1.00 - 0.01); e1 and e3 are not;
    # wait: e3..e7 = 5 -> stops at epoch index 7, restores the weights of e2 (3 steps per epoch -> w = 9)
    assert es.stopped_epoch == 7 and len(History.history["val_loss"]) == 8 and model.steps == 24
    assert np.array_equal(model.w, np.full(3, 9.0))
    # ModelCheckpoint(save_best_only) has NO min_delta: every strict improvement is saved -> e0, e1, e2, e3
    assert [p for p, _ in model.saved] == [weightFile] * 4
    assert [float(w[0]) for _, w in model.saved] == [3.0, 6.0, 9.0, 12.0]
    rows = list(csv.DictReader(open(logFile)))
    assert len(rows) == 8 and rows[0]["epoch"] == "0"
    assert list(rows[0].keys()) == ["epoch"] + sorted(History.history.keys())  # Keras' column order
    assert float(rows[2]["val_loss"]) == pytest.approx(0.96)
    assert float(rows[0]["loss"]) == pytest.approx(1.0 + 0.05, rel=1e-6)  # weighted sum + l2 penalty, mean over the epoch's steps


def test_no_restore_without_early_stop_and_patience_counts():
    es = EarlyStopping(monitor="val_loss", min_delta=0.0, patience=3, restore_best_weights=True)
    model = Scripted([1.0, 0.9, 0.95, 0.93, 0.8])
    h = model.fit(_gen(), steps_per_epoch=1, validation_data=_gen(), validation_steps=1, epochs=5, verbose=0, callbacks=[es])
    assert len(h.history["loss"]) == 5 and es.stopped_epoch == 0
    assert np.array_equal(model.w, np.full(3, 5.0))  # tf.keras 2.x restores only when it stops the run


def test_checkpoint_formats_epoch_and_mode(tmp_path):
    model = Scripted([0.5, 0.4, 0.6])
    mcp = ModelCheckpoint(str(tmp_path / "w.{epoch:02d}-{val_loss:.2f}.h5"), save_weights_only=True)
    model.fit(_gen(), steps_per_epoch=1, validation_data=_gen(), validation_steps=1, epochs=3, verbose=0, callbacks=[mcp])
    assert [p.split("/")[-1] for p, _ in model.saved] == ["w.01-0.50.h5", "w.02-0.40.h5", "w.03-0.60.h5"]
    acc = ModelCheckpoint("x", monitor="val_3C_accuracy", save_best_only=True)
    assert acc.monitor_op is np.greater  # mode='auto' on an accuracy maximises
    with pytest.raises(ValueError):
        ModelCheckpoint("x", save_freq=100)
    with pytest.raises(TypeError):
        ModelCheckpoint("x", period=2)


def test_fit_rejects_what_it_does_not_implement():
    model = Scripted([1.0])
    with pytest.raises(TypeError, match="callbacks"):
        model.fit(_gen(), steps_per_epoch=1, epochs=1, verbose=0, callbacks=[object()])
    with pytest.raises(TypeError, match="unsupported"):
        model.fit(_gen(), steps_per_epoch=1, epochs=1, verbose=0, class_weights={0: 1.0})
    with pytest.raises(ValueError):
        model.fit(_gen(), epochs=1, verbose=0)  # a generator needs steps_per_epoch
    model.fit(_gen(), steps_per_epoch=1, epochs=1, verbose=0, workers=1, use_multiprocessing=False)  # harmless Keras arguments


def test_csv_logger_append_and_custom_callback(tmp_path):
    log = str(tmp_path / "log.csv")
    seen = []

    class Spy(Callback):
        def on_epoch_end(self, epoch, logs=None):
            seen.append((epoch, round(logs["val_loss"], 3)))

    m = Scripted([0.9, 0.8])
    m.fit(_gen(), steps_per_epoch=1, validation_data=_gen(), validation_steps=1, epochs=2, verbose=0, callbacks=[CSVLogger(log), Spy()])
    m2 = Scripted([0.7])
    m2.fit(_gen(), steps_per_epoch=1, validation_data=_gen(), validation_steps=1, epochs=3, initial_epoch=2, verbose=0,
           callbacks=[CSVLogger(log, append=True)])
    rows = list(csv.DictReader(open(log)))
    assert [r["epoch"] for r in rows] == ["0", "1", "2"] and seen == [(0, 0.9), (1, 0.8)]
    # the DAFx driver resumes by counting the lines of this file (DAFx12...:533-545): header + one line per epoch
    assert sum(1 for _ in open(log)) == 4


def test_optimizer_descriptions_follow_keras():
    from sm_hpss_mtl_amd import optimizers
    sch = optimizers.ExponentialDecay(0.002, decay_steps=30, decay_rate=0.1)
    sgd = optimizers.SGD(learning_rate=sch, clipnorm=1, momentum=0.9)  # lib/proposed_architectures.py:156-158
    assert sgd.lr_at(0) == 0.002 and sgd.lr_at(30) == pytest.approx(0.0002) and sgd.clipnorm == 1.0 and sgd.momentum == 0.9
    adam = optimizers.Adam(lr=1e-4)  # the reference's spelling (Proposed_Work_Results.py:381)
    assert adam.lr_at(5) == 1e-4 and (adam.beta_1, adam.beta_2, adam.epsilon) == (0.9, 0.999, 1e-7)
    nadam = optimizers.Nadam(learning_rate=0.002)  # DAFx12...:524-526
    assert nadam.kind == "nadam" and nadam.clipnorm is None
    with pytest.raises(TypeError):
        optimizers.SGD(decay=1e-6)
