"""Training callbacks the reference wires into every trainer (src/trainers.py:934-1068): Keras' ModelCheckpoint, CSVLogger,
EarlyStopping and ReduceLROnPlateau, plus the reference's own StepwiseLearningRateScheduler
(src/callbacks/stepwise_learning_rate_scheduler.py) and OptimizerCheckpoint (src/callbacks/optimizer_checkpoint.py).

Host-side bookkeeping only.  What they change on the optimizer (lr, beta_2) reaches a captured training step through the
optimizer's device hyper-parameter block (keras_extensions/extended_optimizers.py), not through a re-capture.
"""
from __future__ import annotations

import csv
import json
import os
from typing import Callable, List, Optional

import numpy as np


class Callback(object):
    """The subset of keras.callbacks.Callback the engine drives."""

    model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_begin(self, batch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class CallbackList(object):
    def __init__(self, callbacks, model):
        self.callbacks: List[Callback] = list(callbacks or [])
        for c in self.callbacks:
            c.set_model(model)

    def __getattr__(self, name):
        if not name.startswith("on_"):
            raise AttributeError(name)

        def call(*a, **kw):
            for c in self.callbacks:
                getattr(c, name)(*a, **kw)
        return call


def _mode_op(mode, monitor):
    """Keras' 'auto' rule: metrics whose name contains 'acc' are maximised, everything else minimised."""
    if mode == "max" or (mode not in ("min", "max") and ("acc" in monitor or monitor.startswith("fmeasure"))):
        return np.greater, -np.inf
    return np.less, np.inf


class StepwiseLearningRateScheduler(Callback):
    """Sets optimizer.lr (and, for the package's Adam, beta_2) before every batch from functions of the global step index.

    The reference applies the beta_2 schedule only `if isinstance(optimizer, keras.optimizers.Adam)`
    (stepwise_learning_rate_scheduler.py:62), which its own Adam (a subclass of Optimizer) never satisfies, so beta_2 was
    never rescheduled there.  The default (`apply_b2_schedule=False`) reproduces that: beta_2 stays what the optimizer was
    built with; True applies the schedule as written in the configuration (trainer config key
    stepwise_learning_rate_scheduler.apply_b2_schedule)."""

    def __init__(self, lr_schedule: Callable[[int], float], initial_step=0, b2_schedule=None, last_scheduled_step=None,
                 verbose=False, apply_b2_schedule=False):
        assert lr_schedule is not None
        self.lr_schedule, self.b2_schedule = lr_schedule, b2_schedule
        self.step_index = int(initial_step)
        self.last_scheduled_step = int(last_scheduled_step) if last_scheduled_step is not None else None
        self.verbose = verbose
        self.apply_b2_schedule = apply_b2_schedule

    def on_batch_begin(self, batch, logs=None):
        opt = self.model.optimizer
        if not hasattr(opt, 'lr'):
            raise ValueError('Optimizer must have a "lr" attribute.')
        if self.last_scheduled_step is not None and self.step_index > self.last_scheduled_step:
            return
        lr = self.lr_schedule(self.step_index)
        if not isinstance(lr, (float, np.float32, np.float64)):
            raise ValueError('The output of the "schedule" function should be float.')
        opt.lr = float(lr)
        if self.b2_schedule and self.apply_b2_schedule and hasattr(opt, "beta_2"):
            opt.beta_2 = float(self.b2_schedule(self.step_index))
        if self.verbose:
            print('StepwiseLearningRateScheduler: Step {}, setting learning rate to: {}'.format(self.step_index, lr))
        self.step_index += 1


class OptimizerCheckpoint(Callback):
    """Rewrites the optimizer configuration (JSON) at the beginning of every epoch."""

    def __init__(self, log_file_path, enabled=True):
        self.log_file_path, self.enabled = log_file_path, enabled

    def on_epoch_begin(self, epoch, logs=None):
        if not self.enabled or not self.log_file_path:
            return
        os.makedirs(os.path.dirname(self.log_file_path) or ".", exist_ok=True)
        with open(self.log_file_path, 'w') as f:
            f.write(json.dumps(self.model.optimizer.get_config()))


class ModelCheckpoint(Callback):
    """keras.callbacks.ModelCheckpoint: `filepath` may name {epoch} and any key of the epoch's logs ({val_loss}).
    save_weights_only=False stores the optimizer state beside the weights (ExtendedModel.save)."""

    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False, mode='auto',
                 period=1, enabled=True):
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only, self.period = save_best_only, save_weights_only, int(period)
        self.op, self.best = _mode_op(mode, monitor)
        self.epochs_since_last_save = 0
        self.enabled = enabled      # data parallel: one writer per file (rank 0)
        self.last_path = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since_last_save += 1
        if self.epochs_since_last_save < self.period:
            return
        self.epochs_since_last_save = 0
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None or not self.op(cur, self.best):
                return
            self.best = cur
        fields = dict(logs)
        for k, v in logs.items():      # a run without validation data still names its files: val_x falls back to x
            fields.setdefault("val_" + k, v)
        path = self.filepath.format(epoch=epoch, **{k: v for k, v in fields.items() if np.ndim(v) == 0})
        self.last_path = path
        if not self.enabled:
            return
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if self.save_weights_only:
            self.model.save_weights(path, overwrite=True)
        else:
            self.model.save(path, overwrite=True)
        if self.verbose:
            print('Epoch {:05d}: saving model to {}'.format(epoch, path))


class CSVLogger(Callback):
    def __init__(self, filename, separator=',', append=False, enabled=True):
        self.filename, self.sep, self.append, self.enabled = filename, separator, append, enabled
        self.keys = None
        self._file = self._writer = None

    def on_train_begin(self, logs=None):
        if not self.enabled:
            return
        os.makedirs(os.path.dirname(self.filename) or ".", exist_ok=True)
        self._file = open(self.filename, 'a' if self.append else 'w', newline='')

    def on_epoch_end(self, epoch, logs=None):
        if self._file is None:
            return
        logs = {k: v for k, v in (logs or {}).items() if np.ndim(v) == 0}
        if self.keys is None:
            self.keys = sorted(logs.keys())
            self._writer = csv.DictWriter(self._file, fieldnames=['epoch'] + self.keys, delimiter=self.sep)
            self._writer.writeheader()
        row = {'epoch': epoch}
        row.update({k: logs.get(k, 'NA') for k in self.keys})
        self._writer.writerow(row)
        self._file.flush()

    def on_train_end(self, logs=None):
        if self._file is not None:
            self._file.close()
            self._file = None


class EarlyStopping(Callback):
    def __init__(self, monitor='val_loss', min_delta=0.0, patience=0, verbose=0, mode='auto'):
        self.monitor, self.patience, self.verbose = monitor, int(patience), verbose
        self.op, self.best = _mode_op(mode, monitor)
        self.min_delta = abs(min_delta) * (1 if self.op is np.greater else -1)
        self.wait = 0
        self.stopped_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.best = np.inf if self.op is np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.op(cur - self.min_delta, self.best):
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True


class ReduceLROnPlateau(Callback):
    def __init__(self, monitor='val_loss', factor=0.1, patience=10, verbose=0, mode='auto', epsilon=1e-4, cooldown=0, min_lr=0):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, int(patience), verbose
        self.epsilon, self.cooldown, self.min_lr = epsilon, int(cooldown), min_lr
        op, self.best = _mode_op(mode, monitor)
        self.monitor_op = (lambda a, b: np.less(a, b - epsilon)) if op is np.less else (lambda a, b: np.greater(a, b + epsilon))
        self.cooldown_counter = 0
        self.wait = 0

    def on_epoch_end(self, epoch, logs=None):
        logs = logs if logs is not None else {}
        opt = self.model.optimizer
        logs['lr'] = opt.lr
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self.monitor_op(cur, self.best):
            self.best, self.wait = cur, 0
        elif self.cooldown_counter <= 0:
            if self.wait >= self.patience:
                if opt.lr > self.min_lr + 1e-4 * max(self.min_lr, 1e-12):
                    opt.lr = max(opt.lr * self.factor, self.min_lr)
                    self.cooldown_counter = self.cooldown
                    self.wait = 0
            self.wait += 1
