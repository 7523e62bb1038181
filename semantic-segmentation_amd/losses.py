"""Loss functions with the call shapes of the reference's src/losses.py, computed by libssmi.so.

Tensors are float32/int32 CUDA (ROCm) torch tensors in NHWC; each function returns a 0-d CUDA float tensor.
The fused kernel also produces the logits gradient; `last_gradient()` hands it to callers that drive their own
backward pass (the model engine does this internally).
"""
from __future__ import annotations

from enum import Enum
from typing import Optional

import numpy as np
import torch

from ._lib import lib


class ModelLambdaLossType(Enum):
    """src/losses.py:11-22."""
    NONE = 0
    SEGMENTATION_CATEGORICAL_CROSS_ENTROPY = 1
    SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER = 2
    SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL = 3
    SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL = 4
    CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY = 5
    CLASSIFICATION_SEMI_SUPERVISED_MEAN_TEACHER = 6


_last = {}


def last_gradient() -> Optional[torch.Tensor]:
    """d(loss)/d(logits) of the most recent loss call on this process."""
    return _last.get("dlogits")


def _xent(logits: torch.Tensor, labels: torch.Tensor, weights: torch.Tensor, want_grad=True):
    if logits.dim() != 4:
        raise ValueError("y_pred must have rank 4 (BxHxWxC), got rank {}".format(logits.dim()))
    if labels.dim() != 3 or weights.dim() != 3:
        raise ValueError("y_true and weights must have rank 3 (BxHxW)")
    if tuple(labels.shape) != tuple(logits.shape[:3]) or tuple(weights.shape) != tuple(labels.shape):
        raise ValueError("shape mismatch: logits {}, labels {}, weights {}".format(
            tuple(logits.shape), tuple(labels.shape), tuple(weights.shape)))
    if not logits.is_cuda:
        raise RuntimeError("the loss runs on the GPU only (libssmi.so); got a CPU tensor")
    L = lib()
    logits = logits.contiguous().float()
    labels = labels.contiguous().to(torch.int32)
    weights = weights.contiguous().float()
    npix = labels.numel()
    C = logits.shape[-1]
    ws = torch.empty(L.softmax_xent_workspace_bytes(npix) // 4 + 4, dtype=torch.float32, device=logits.device)
    loss = torch.zeros((), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    st = torch.cuda.current_stream().cuda_stream
    L.softmax_xent_fwd_bwd(logits.data_ptr(), labels.data_ptr(), weights.data_ptr(), npix, C, 1.0,
                           None if dl is None else dl.data_ptr(), loss.data_ptr(), None, 0, 0, None, None,
                           ws.data_ptr(), ws.numel() * 4, st)
    _last["dlogits"] = dl
    return loss


def dummy_loss(y_true, y_pred):
    """src/losses.py:222-236: passes the lambda-loss layer's output through."""
    return y_pred


def _preprocess_segmentation_lambda_loss_args(args, num_expected_args):
    """src/losses.py:310-331."""
    if len(args) != num_expected_args:
        raise ValueError('Expected {} arguments, got: {} ({})'.format(num_expected_args, len(args), args))
    y_pred, y_true, weights, num_unlabeled = args[0], args[1], args[2], args[3]
    if y_pred.dim() != 4 or y_true.dim() != 3 or weights.dim() != y_true.dim() or num_unlabeled.dim() != 2:
        raise ValueError("rank mismatch: y_pred 4, y_true 3, weights 3, num_unlabeled 2 expected")
    nu = int(num_unlabeled.reshape(-1)[0].item())
    num_labeled = y_true.shape[0] - nu
    return y_pred, y_true, weights[0:num_labeled], nu, num_labeled


def segmentation_categorical_cross_entropy_lambda_loss(args):
    """src/losses.py:341-374.  args = [logits BxHxWxC, labels BxHxW int32, weights BxHxW, num_unlabeled Bx1 int32].
    As in the reference the labels/logits are not sliced, so num_unlabeled must be 0 for the shapes to agree."""
    y_pred, y_true, weights, num_unlabeled, num_labeled = _preprocess_segmentation_lambda_loss_args(args, 4)
    return _xent(y_pred, y_true, weights)


def _consistency(y_pred, mt_pred, coefficient: float, dlogits, npix_accumulate: int, loss: torch.Tensor):
    """Adds coefficient * mean((softmax(mt_pred) - softmax(y_pred))^2) (src/losses.py:195-215) to `loss` and its
    gradient to `dlogits`."""
    if tuple(mt_pred.shape) != tuple(y_pred.shape):
        raise ValueError("mt_pred {} and y_pred {} must have the same shape".format(tuple(mt_pred.shape), tuple(y_pred.shape)))
    L = lib()
    y_pred = y_pred.contiguous().float()
    mt_pred = mt_pred.contiguous().float()
    C = y_pred.shape[-1]
    npix = y_pred.numel() // C
    ws = torch.empty(L.mt_consistency_workspace_bytes(npix) // 4 + 4, dtype=torch.float32, device=y_pred.device)
    coef = torch.full((1,), float(coefficient), dtype=torch.float32, device=y_pred.device)
    lbuf = loss.reshape(1)
    st = torch.cuda.current_stream().cuda_stream
    L.mt_consistency_fwd_bwd(y_pred.data_ptr(), mt_pred.data_ptr(), npix, C, coef.data_ptr(), 1.0, npix_accumulate,
                             None if dlogits is None else dlogits.data_ptr(), lbuf.data_ptr(), 1, None, ws.data_ptr(),
                             ws.numel() * 4, st)
    return loss


def segmentation_mean_teacher_lambda_loss(args):
    """src/losses.py:377-436.  args = [logits BxHxWxC, labels BxHxW int32, weights BxHxW, num_unlabeled Bx1 int32,
    mt_pred BxHxWxC (teacher logits), cons_coefficient Bx1].  Classification cost over the first B - num_unlabeled rows,
    consistency cost over every row."""
    return _semi_supervised(args, 6, True, False)


SUPERPIXEL_EPSILON = 10e-7      # src/settings.py:6


def _superpixel(y_pred_unlabeled, y_true_unlabeled, coefficient: float, dlogits_unlabeled, accumulate: bool, loss: torch.Tensor):
    """Adds the superpixel cost of the unlabeled rows (src/losses.py:108-192) to `loss` and its gradient to the
    unlabeled rows of the logits gradient (a contiguous view)."""
    L = lib()
    nu, h, w, C = y_pred_unlabeled.shape
    mask = y_true_unlabeled.contiguous().to(torch.int32)
    ws = torch.empty(L.superpixel_workspace_bytes(nu * h * w, C) // 4 + 4, dtype=torch.float32, device=y_pred_unlabeled.device)
    coef = torch.full((1,), float(coefficient), dtype=torch.float32, device=y_pred_unlabeled.device)
    st = torch.cuda.current_stream().cuda_stream
    L.superpixel_cost_fwd_bwd(y_pred_unlabeled.data_ptr(), mask.data_ptr(), nu, h, w, C, coef.data_ptr(), 1.0, SUPERPIXEL_EPSILON,
                              None if dlogits_unlabeled is None else dlogits_unlabeled.data_ptr(), int(accumulate),
                              loss.reshape(1).data_ptr(), 1, None, ws.data_ptr(), ws.numel() * 4, st)
    return loss


def _semi_supervised(args, n_args, with_mt, with_sp):
    y_pred, y_true, weights_labeled, num_unlabeled, num_labeled = _preprocess_segmentation_lambda_loss_args(args, n_args)
    if num_labeled <= 0:
        raise ValueError("the classification cost needs at least one labeled row")
    for extra in args[4:]:
        if extra.dim() not in (2, y_pred.dim()):
            raise ValueError("rank mismatch in the semi-supervised loss inputs")
    y_pred = y_pred.contiguous().float()
    loss = _xent(y_pred[0:num_labeled], y_true[0:num_labeled], weights_labeled)
    dl = torch.zeros_like(y_pred)
    dl[0:num_labeled].copy_(_last["dlogits"])
    npl = num_labeled * y_pred.shape[1] * y_pred.shape[2]
    if with_mt:
        mt_pred, coef = args[4], args[5]
        if mt_pred.dim() != y_pred.dim() or coef.dim() != 2:
            raise ValueError("rank mismatch: mt_pred must have the rank of y_pred, cons_coefficient rank 2")
        _consistency(y_pred, mt_pred, float(coef.reshape(-1)[0].item()), dl, npl, loss)
    if with_sp and num_unlabeled > 0:
        sp_coef = args[-1]
        if sp_coef.dim() != 2:
            raise ValueError("rank mismatch: superpixel_consistency_cost_coefficient must have rank 2")
        _superpixel(y_pred[num_labeled:], y_true[num_labeled:], float(sp_coef.reshape(-1)[0].item()), dl[num_labeled:],
                    with_mt, loss)
    _last["dlogits"] = dl
    return loss


def segmentation_superpixel_lambda_loss(args):
    """src/losses.py:439-492.  args = [logits, labels (superpixel masks on the unlabeled rows), weights, num_unlabeled,
    superpixel_consistency_cost_coefficient Bx1]."""
    if len(args) != 5:
        raise ValueError('Expected 5 values (y_pred, y_true, weights, num_unlabeled, unlabeled_cost_coefficient), got: {} ({})'.format(len(args), args))
    return _semi_supervised(args, 5, False, True)


def segmentation_mean_teacher_superpixel_lambda_loss(args):
    """src/losses.py:495-572.  args = [logits, labels, weights, num_unlabeled, mt_pred, consistency_coefficient Bx1,
    superpixel_consistency_cost_coefficient Bx1]."""
    return _semi_supervised(args, 7, True, True)


def _classification_xent(y_pred, y_true, class_weights):
    if y_pred.dim() != 2 or y_true.dim() != 2 or class_weights.dim() != 2:
        raise ValueError("y_pred, y_true and the class weights must have rank 2 (batch x classes)")
    if tuple(y_true.shape) != tuple(y_pred.shape) or tuple(class_weights.shape) != tuple(y_pred.shape):
        raise ValueError("shape mismatch: logits {}, labels {}, weights {}".format(
            tuple(y_pred.shape), tuple(y_true.shape), tuple(class_weights.shape)))
    if not y_pred.is_cuda:
        raise RuntimeError("the loss runs on the GPU only (libssmi.so); got a CPU tensor")
    L = lib()
    z, y, w = y_pred.contiguous().float(), y_true.contiguous().float(), class_weights.contiguous().float()
    loss = torch.zeros((), dtype=torch.float32, device=z.device)
    dl = torch.empty_like(z)
    L.classification_xent_fwd_bwd(z.data_ptr(), y.data_ptr(), w.data_ptr(), z.shape[0], z.shape[1], SUPERPIXEL_EPSILON, 1.0,
                                  dl.data_ptr(), loss.data_ptr(), torch.cuda.current_stream().cuda_stream)
    _last["dlogits"] = dl
    return loss


def classification_weighted_categorical_crossentropy_loss(class_weights):
    """src/losses.py:603-609: -mean_n sum_c y_true * log(clip(softmax(y_pred), 1e-6, 1)) * class_weights[c]."""
    if class_weights is None:
        raise ValueError('Class weights is None. Use a numpy array of ones instead of None.')
    cw_host = np.asarray(class_weights, dtype=np.float32)

    def loss(y_true, y_pred):
        cw = torch.from_numpy(cw_host).to(y_pred.device).expand_as(y_pred)
        return _classification_xent(y_pred, y_true, cw)

    return loss


def classification_categorical_crossentropy_lambda_loss(args):
    """src/losses.py:636-652.  args = [logits BxC, labels BxC float, class weights BxC, num_unlabeled Bx1 int32].  As in
    the reference only the weights are sliced to the labeled rows, so num_unlabeled must be 0 for the shapes to agree."""
    if len(args) != 4:
        raise ValueError('Expected 4 arguments, got: {} ({})'.format(len(args), args))
    y_pred, y_true, weights, num_unlabeled = args
    if num_unlabeled.dim() != 2:
        raise ValueError("num_unlabeled must have rank 2")
    if int(num_unlabeled.reshape(-1)[0].item()) != 0:
        raise ValueError("the classification lambda loss requires num_unlabeled == 0 (shape mismatch in the reference)")
    return _classification_xent(y_pred, y_true, weights)


def segmentation_sparse_weighted_categorical_cross_entropy(class_weights):
    """src/losses.py:334-338 (+268-307): y_true BxHxWx1, per-pixel weights = class_weights[y_true]."""
    if class_weights is None:
        raise ValueError('Class weights is None. Use a numpy array of ones instead of None.')
    cw_host = np.asarray(class_weights, dtype=np.float32)

    def loss(y_true, y_pred):
        if y_pred.dim() != 4 or y_true.dim() != 4:
            raise ValueError("y_true and y_pred must have rank 4")
        lab = y_true.squeeze(-1).to(torch.int64)
        cw = torch.from_numpy(cw_host).to(y_pred.device)
        return _xent(y_pred, lab.to(torch.int32), cw[lab])

    return loss
