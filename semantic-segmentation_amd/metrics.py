"""Streaming segmentation metrics with the factory signatures of the reference's src/metrics.py.

Every factory returns a function carrying the reference's attributes (`streaming`, `cfm`, `hide_from_progbar`,
`exclude_from_callbacks`, `reset_op`).  All four metrics share ONE int64 confusion matrix that the fused loss kernel
updates on the device (rows = labels, columns = predictions, ignored labels skipped); accuracy / mean-per-class
accuracy / mean IoU are derived from it on the host exactly like tf.metrics.* derive them from their own totals.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from ._lib import lib

_CFM_DTYPE = np.int64   # src/metrics.py:6


def ignore_mask_bits(ignore_classes: Optional[Sequence[int]]) -> int:
    bits = 0
    for c in (ignore_classes or ()):
        if not 0 <= int(c) < 64:
            raise ValueError("ignore class {} outside [0, 64)".format(c))
        bits |= 1 << int(c)
    return bits


def metrics_from_confusion_matrix(cm: np.ndarray, ignore_classes=None) -> dict:
    """Host-side derivations (ssmi_metrics_from_cm): tf.metrics.accuracy / mean_per_class_accuracy / mean_iou
    (src/metrics.py:284,310,337) and the reference's numpy script definitions
    (src/scripts/calculate_metrics_from_cfms.py:12-53)."""
    cm = np.ascontiguousarray(cm, dtype=np.int64)
    n = cm.shape[0]
    out = (C.c_double * 6)()
    lib().metrics_from_cm(cm.ctypes.data_as(C.c_void_p), n, ignore_mask_bits(ignore_classes), out)
    return dict(accuracy=out[0], mpca=out[1], miou=out[2], miou_all_classes=out[3], script_mpca=out[4], script_miou=out[5])


class StreamingConfusionMatrix:
    """Device-resident epoch-cumulative confusion matrix (src/metrics.py:89-130)."""

    def __init__(self, num_classes: int, ignore_classes=None, device="cuda"):
        self.num_classes = num_classes
        self.ignore_classes = list(ignore_classes or [])
        self.ignore_bits = ignore_mask_bits(self.ignore_classes)
        self.total = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=device)

    def reset(self):
        self.total.zero_()

    def update(self, labels: torch.Tensor, logits: torch.Tensor):
        """Adds one batch (labels [..] int32, logits [.., C]) and returns the cumulative matrix tensor."""
        L = lib()
        logits = logits.contiguous().float()
        labels = labels.contiguous().to(torch.int32)
        npix = labels.numel()
        w = torch.zeros(npix, dtype=torch.float32, device=logits.device)
        ws = torch.empty(L.softmax_xent_workspace_bytes(npix) // 4 + 4, dtype=torch.float32, device=logits.device)
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        L.softmax_xent_fwd_bwd(logits.data_ptr(), labels.data_ptr(), w.data_ptr(), npix, self.num_classes, 0.0, None,
                               loss.data_ptr(), self.total.data_ptr(), npix, self.ignore_bits, None, None, ws.data_ptr(),
                               ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
        return self.total

    def numpy(self) -> np.ndarray:
        return self.total.cpu().numpy().astype(_CFM_DTYPE)


def _make(name, num_classes, num_unlabeled, ignore_classes, **attrs):
    def fn(y_true, y_pred):
        """Stand-alone use: y_true BxHxWx1, y_pred BxHxWxC (CUDA tensors).  Updates the metric's own streaming
        confusion matrix and returns the metric's current value."""
        if y_true.dim() != 4 or y_pred.dim() != 4:
            raise ValueError("y_true must be BxHxWx1 and y_pred BxHxWxC")
        if fn.state is None:
            fn.state = StreamingConfusionMatrix(fn.num_classes or y_pred.shape[-1], fn.ignore_classes, y_pred.device)
        nu = int(fn.num_unlabeled) if not torch.is_tensor(fn.num_unlabeled) else int(fn.num_unlabeled.reshape(-1)[0])
        nl = y_true.shape[0] - nu
        fn.state.update(y_true[0:nl].squeeze(-1), y_pred[0:nl])
        return fn.value_from_cm(fn.state.numpy())

    def reset_op():
        if fn.state is not None:
            fn.state.reset()

    def value_from_cm(cm):
        if name == "cfm":
            return cm
        m = metrics_from_confusion_matrix(cm, fn.ignore_classes)
        return {"acc": m["accuracy"], "mpca": m["mpca"], "miou": m["miou"]}[name]

    fn.__name__ = name
    fn.num_classes, fn.num_unlabeled, fn.ignore_classes = num_classes, num_unlabeled, list(ignore_classes or [])
    fn.state = None
    fn.reset_op = reset_op
    fn.value_from_cm = value_from_cm
    fn.streaming = True
    for k, v in attrs.items():
        setattr(fn, k, v)
    return fn


def segmentation_accuracy(num_unlabeled, ignore_classes=None):
    """src/metrics.py:276-298."""
    return _make("acc", None, num_unlabeled, ignore_classes)


def segmentation_mean_per_class_accuracy(num_classes, num_unlabeled, ignore_classes=None):
    """src/metrics.py:301-325."""
    return _make("mpca", num_classes, num_unlabeled, ignore_classes)


def segmentation_mean_iou(num_classes, num_unlabeled, ignore_classes=None):
    """src/metrics.py:328-352."""
    return _make("miou", num_classes, num_unlabeled, ignore_classes)


def segmentation_confusion_matrix(num_classes, num_unlabeled, ignore_classes=None):
    """src/metrics.py:355-378."""
    return _make("cfm", num_classes, num_unlabeled, ignore_classes, hide_from_progbar=True, exclude_from_callbacks=True,
                 cfm=True)
