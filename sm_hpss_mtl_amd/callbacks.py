"""The three Keras callbacks the reference hands to `model.fit(..., callbacks=[csv_logger, es, mcp])`
(Proposed_Work_Results.py:276-278, 298-307; DAFx12_Speech_Music_Detection_B3_MTL_v2.py transfer_learn_model), with the
constructor arguments used there and tf.keras 2.x semantics:

    es  = EarlyStopping(monitor='val_loss', mode='auto', verbose=1, restore_best_weights=True, min_delta=0.01, patience=5)
    mcp = ModelCheckpoint(weightFile, monitor='val_loss', verbose=0, save_best_only=True, save_weights_only=True,
                          mode='auto', save_freq='epoch')
    csv_logger = CSVLogger(logFile)

EarlyStopping and ModelCheckpoint keep SEPARATE bests: the checkpoint is written on every strict improvement of the
monitored value, early stopping counts an epoch as an improvement only when it beats its best by more than min_delta;
the best weights are restored when (and only when) training is stopped early, as in tf.keras 2.x.
"""
from __future__ import annotations

import csv
import os

import numpy as np


class Callback:
    """`is_writer`: data-parallel training runs the same callbacks on every rank with identical logs (fit all-reduces them), so
    every rank takes the same decisions; only the writer (rank 0) touches the file system."""

    def __init__(self):
        self.model = None
        self.is_writer = True

    def set_model(self, model):
        self.model = model

    def set_writer(self, is_writer):
        self.is_writer = bool(is_writer)

    def on_train_begin(self, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass


def _monitor_op(mode, monitor):
    if mode not in ("auto", "min", "max"):
        raise ValueError("mode must be 'auto', 'min' or 'max', got %r" % (mode,))
    if mode == "max" or (mode == "auto" and ("acc" in monitor or monitor.startswith("fmeasure"))):
        return np.greater, -np.inf
    return np.less, np.inf


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto", baseline=None,
                 restore_best_weights=False):
        super().__init__()
        self.monitor, self.patience, self.verbose, self.baseline = monitor, int(patience), verbose, baseline
        self.restore_best_weights = bool(restore_best_weights)
        self.monitor_op, self._worst = _monitor_op(mode, monitor)
        self.min_delta = abs(float(min_delta)) * (1.0 if self.monitor_op is np.greater else -1.0)
        self.wait, self.stopped_epoch, self.best, self.best_weights = 0, 0, self._worst, None

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch, self.best_weights = 0, 0, None
        self.best = self.baseline if self.baseline is not None else self._worst

    def on_epoch_end(self, epoch, logs=None):
        current = (logs or {}).get(self.monitor)
        if current is None:  # Keras warns and goes on
            return
        if self.monitor_op(current - self.min_delta, self.best):
            self.best, self.wait = current, 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.restore_best_weights and self.best_weights is not None:
                    if self.verbose:
                        print("Restoring model weights from the end of the best epoch.")
                    self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose:
            print("Epoch %05d: early stopping" % (self.stopped_epoch + 1))


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False, mode="auto",
                 save_freq="epoch", **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError("ModelCheckpoint: unsupported arguments %s" % sorted(kwargs))
        if save_freq != "epoch":
            raise ValueError("ModelCheckpoint: only save_freq='epoch' is supported (the reference's setting)")
        self.filepath, self.monitor, self.verbose = str(filepath), monitor, verbose
        self.save_best_only, self.save_weights_only = bool(save_best_only), bool(save_weights_only)
        self.monitor_op, self.best = _monitor_op(mode, monitor)

    def _save(self, path):
        if not self.is_writer:  # data parallel: replicas are identical, rank 0 writes the one file
            return
        self.model.save_weights(path)
        if not self.save_weights_only:  # whole-model save: the architecture goes next to the weights
            with open(os.path.splitext(path)[0] + ".json", "w") as f:
                f.write(self.model.to_json())

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if not self.save_best_only:
            return self._save(path)
        current = logs.get(self.monitor)
        if current is None:
            return
        if self.monitor_op(current, self.best):
            if self.verbose:
                print("Epoch %05d: %s improved from %0.5f to %0.5f, saving model to %s" % (epoch + 1, self.monitor, self.best, current, path))
            self.best = current
            self._save(path)


class CSVLogger(Callback):
    """One row per epoch: `epoch` then the log keys in sorted order (Keras' column order), flushed every epoch."""

    def __init__(self, filename, separator=",", append=False):
        super().__init__()
        self.filename, self.sep, self.append = str(filename), separator, bool(append)
        self.keys, self._file, self._writer = None, None, None

    def on_train_begin(self, logs=None):
        if not self.is_writer:
            return
        self._append_header = not (self.append and os.path.exists(self.filename) and os.path.getsize(self.filename) > 0)
        self._file = open(self.filename, "a" if self.append else "w", newline="")

    def on_epoch_end(self, epoch, logs=None):
        if not self.is_writer:
            return
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs.keys())
        if self._writer is None:
            self._writer = csv.DictWriter(self._file, fieldnames=["epoch"] + self.keys, delimiter=self.sep)
            if self._append_header:
                self._writer.writeheader()
        row = {"epoch": epoch}
        row.update((k, logs.get(k, "NA")) for k in self.keys)
        self._writer.writerow(row)
        self._file.flush()

    def on_train_end(self, logs=None):
        if self._file is not None:
            self._file.close()
            self._file, self._writer = None, None
