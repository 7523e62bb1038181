"""Adam / SGD with per-weight learning-rate scalers (reference: src/keras_extensions/extended_optimizers.py).

The optimizer owns its slot arenas (m, v / momentum) shaped like the model's flat parameter arena and applies one
fused kernel per step.  `iterations` lives on the device so that a captured HIP graph can be replayed.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .._lib import lib


class Optimizer:
    """Hyper-parameters (lr, decay, betas / momentum) are plain attributes on the host, as in Keras where callbacks set
    them between steps (src/callbacks/stepwise_learning_rate_scheduler.py:59-64).  The update kernels read them from a
    small device block, `hyper`, which `flush_hyper` rewrites whenever an attribute (or the gradient scale) has
    changed, so a step captured into a HIP graph keeps following them."""

    _HYPER = ()      # attribute names whose change must reach the device block

    def __init__(self, lr_scalers=None):
        self.lr_scalers: Dict[int, float] = dict(lr_scalers or {})
        self._bound = False
        self._hyper_sent = None

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in self._HYPER:
            object.__setattr__(self, "_hyper_sent", None)

    def _hyper_values(self, grad_scale):
        raise NotImplementedError

    def flush_hyper(self, grad_scale: float = 1.0):
        """Writes the current hyper-parameters to the device block if they differ from what it holds.  Call outside
        graph capture, on the stream the step runs on (a small host-to-device copy, ordered before the next launch)."""
        vals = tuple(float(v) for v in self._hyper_values(grad_scale))
        if vals != self._hyper_sent:
            self.hyper.copy_(torch.tensor(vals + (0.0,) * (8 - len(vals)), dtype=torch.float32))
            object.__setattr__(self, "_hyper_sent", vals)

    def bind(self, store, frozen_keys=()):
        """Attach to a parameter store (engine.ParamStore).  `frozen_keys`: weights of layers with trainable = False.
        Keras builds `params` from the trainable weights only, so the indices of `lr_scalers`
        (extended_optimizers.py:41,124: `lr_scalers[i]` scales params[i]) count the non-frozen weights in creation
        order; frozen weights get a rate of exactly zero."""
        self.store = store
        dev = store.theta.device
        # 16-byte block: int64 iterations + 8 bytes the library uses for the step's effective learning rate
        self.step_block = torch.zeros(2, dtype=torch.int64, device=dev)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        object.__setattr__(self, "_hyper_sent", None)
        self.lr_scale = None
        frozen = set(frozen_keys)
        if self.lr_scalers or frozen:
            tab = torch.ones_like(store.theta)
            i = 0
            for key in store.trainable_keys():
                s = store.specs[key]
                if key in frozen:
                    tab[s.offset:s.offset + s.size] = 0.0
                    continue
                if i in self.lr_scalers:
                    tab[s.offset:s.offset + s.size] = float(self.lr_scalers[i])
                i += 1
            self.lr_scale = tab
        self._bound = True

    @property
    def iterations(self) -> int:
        return int(self.step_block[0].item())

    def set_iterations(self, v: int):
        self.step_block[0] = int(v)


class Adam(Optimizer):
    """extended_optimizers.py:79-173."""

    _HYPER = ("lr", "beta_1", "beta_2", "epsilon", "decay", "initial_decay")

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0., lr_scalers=None, amsgrad=False, **kwargs):
        super().__init__(lr_scalers)
        self.lr, self.beta_1, self.beta_2, self.decay = float(lr), float(beta_1), float(beta_2), float(decay)
        self.epsilon = 1e-7 if epsilon is None else float(epsilon)   # K.epsilon()
        self.initial_decay = float(decay)
        self.amsgrad = amsgrad

    def bind(self, store, frozen_keys=()):
        super().bind(store, frozen_keys)
        self.m = torch.zeros_like(store.theta)
        self.v = torch.zeros_like(store.theta)
        self.vhat = torch.zeros_like(store.theta) if self.amsgrad else None

    def _hyper_values(self, grad_scale):
        return (self.lr, self.beta_1, self.beta_2, self.epsilon, self.decay if self.initial_decay > 0 else 0.0, grad_scale)

    def apply(self, grad_scale: float = 1.0, stream: Optional[int] = None, capturing: bool = False):
        """One update.  `capturing`: the call is being recorded into a HIP graph; the device block was flushed by the
        caller beforehand (a copy cannot be part of the recording)."""
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        s = self.store
        if not capturing:
            self.flush_hyper(grad_scale)
        lib().adam_step_hp(s.theta.data_ptr(), s.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                           None if self.vhat is None else self.vhat.data_ptr(), s.theta.numel(), self.step_block.data_ptr(),
                           self.hyper.data_ptr(), None if self.lr_scale is None else self.lr_scale.data_ptr(), st)

    @classmethod
    def from_config(cls, config):
        return cls(**{k: (({int(i): v for i, v in val.items()}) if k == "lr_scalers" and val else val)
                      for k, val in config.items()})

    def get_config(self):
        return {'lr': self.lr, 'beta_1': self.beta_1, 'beta_2': self.beta_2, 'decay': self.decay, 'epsilon': self.epsilon,
                'amsgrad': self.amsgrad, 'lr_scalers': self.lr_scalers}

    def slot_tensors(self):
        return {"m": self.m, "v": self.v, **({"vhat": self.vhat} if self.vhat is not None else {})}


class SGD(Optimizer):
    """extended_optimizers.py:8-76."""

    _HYPER = ("lr", "momentum", "decay", "initial_decay", "nesterov")

    def __init__(self, lr=0.01, momentum=0., decay=0., nesterov=False, lr_scalers=None, **kwargs):
        super().__init__(lr_scalers)
        self.lr, self.momentum, self.decay = float(lr), float(momentum), float(decay)
        self.initial_decay = float(decay)
        self.nesterov = bool(nesterov)

    def bind(self, store, frozen_keys=()):
        super().bind(store, frozen_keys)
        self.moments = torch.zeros_like(store.theta)

    def _hyper_values(self, grad_scale):
        return (self.lr, self.momentum, self.decay if self.initial_decay > 0 else 0.0, 1.0 if self.nesterov else 0.0, grad_scale)

    def apply(self, grad_scale: float = 1.0, stream: Optional[int] = None, capturing: bool = False):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        s = self.store
        if not capturing:
            self.flush_hyper(grad_scale)
        lib().sgd_step_hp(s.theta.data_ptr(), s.grad.data_ptr(), self.moments.data_ptr(), s.theta.numel(),
                          self.step_block.data_ptr(), self.hyper.data_ptr(),
                          None if self.lr_scale is None else self.lr_scale.data_ptr(), st)

    @classmethod
    def from_config(cls, config):
        return cls(**{k: (({int(i): v for i, v in val.items()}) if k == "lr_scalers" and val else val)
                      for k, val in config.items()})

    def get_config(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'decay': self.decay, 'nesterov': self.nesterov,
                'lr_scalers': self.lr_scalers}

    def slot_tensors(self):
        return {"moments": self.moments}
