"""Adam / SGD with per-weight learning-rate scalers (reference: src/keras_extensions/extended_optimizers.py).

The optimizer owns its slot arenas (m, v / momentum) shaped like the model's flat parameter arena and applies one
fused kernel per step.  `iterations` lives on the device so that a captured HIP graph can be replayed.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .._lib import lib


class Optimizer:
    def __init__(self, lr_scalers=None):
        self.lr_scalers: Dict[int, float] = dict(lr_scalers or {})
        self._bound = False

    def bind(self, store, frozen_keys=()):
        """Attach to a parameter store (engine.ParamStore).  `frozen_keys`: weights of layers with trainable = False.
        Keras builds `params` from the trainable weights only, so the indices of `lr_scalers`
        (extended_optimizers.py:41,124: `lr_scalers[i]` scales params[i]) count the non-frozen weights in creation
        order; frozen weights get a rate of exactly zero."""
        self.store = store
        dev = store.theta.device
        # 16-byte block: int64 iterations + 8 bytes the library uses for the step's effective learning rate
        self.step_block = torch.zeros(2, dtype=torch.int64, device=dev)
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

    def apply(self, grad_scale: float = 1.0, stream: Optional[int] = None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        s = self.store
        lib().adam_step(s.theta.data_ptr(), s.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                        None if self.vhat is None else self.vhat.data_ptr(), s.theta.numel(), self.step_block.data_ptr(),
                        self.lr, self.beta_1, self.beta_2, self.epsilon, self.decay if self.initial_decay > 0 else 0.0,
                        grad_scale, None if self.lr_scale is None else self.lr_scale.data_ptr(), st)

    def get_config(self):
        return {'lr': self.lr, 'beta_1': self.beta_1, 'beta_2': self.beta_2, 'decay': self.decay, 'epsilon': self.epsilon,
                'amsgrad': self.amsgrad, 'lr_scalers': self.lr_scalers}

    def slot_tensors(self):
        return {"m": self.m, "v": self.v, **({"vhat": self.vhat} if self.vhat is not None else {})}


class SGD(Optimizer):
    """extended_optimizers.py:8-76."""

    def __init__(self, lr=0.01, momentum=0., decay=0., nesterov=False, lr_scalers=None, **kwargs):
        super().__init__(lr_scalers)
        self.lr, self.momentum, self.decay = float(lr), float(momentum), float(decay)
        self.initial_decay = float(decay)
        self.nesterov = bool(nesterov)

    def bind(self, store, frozen_keys=()):
        super().bind(store, frozen_keys)
        self.moments = torch.zeros_like(store.theta)

    def apply(self, grad_scale: float = 1.0, stream: Optional[int] = None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        s = self.store
        lib().sgd_step(s.theta.data_ptr(), s.grad.data_ptr(), self.moments.data_ptr(), s.theta.numel(),
                       self.step_block.data_ptr(), self.lr, self.momentum, self.decay if self.initial_decay > 0 else 0.0,
                       1 if self.nesterov else 0, grad_scale, None if self.lr_scale is None else self.lr_scale.data_ptr(), st)

    def get_config(self):
        return {'lr': self.lr, 'momentum': self.momentum, 'decay': self.decay, 'nesterov': self.nesterov,
                'lr_scalers': self.lr_scalers}

    def slot_tensors(self):
        return {"moments": self.moments}
