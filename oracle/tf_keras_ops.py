"""CPU oracle: restatement of the Keras/TensorFlow operator semantics the hot path relies on.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this file; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path,
and TensorFlow/Keras are not installable here, so these functions restate the published
TF 1.4-1.8 / Keras 2.0.8-2.1.x semantics from the reference's own call sites.  The
hand-computed known-answer cases in tests/test_oracle_kat.py are what pins them.

Substrate: PyTorch CPU (float64 for goldens, float32 for the timed baseline), tensors in
NHWC like the reference (`channels_last`, reference configs/.../config-segmentation-supervised.json:30).
Gradients come from torch autograd over these definitions.

Each function names the reference call site it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# padding arithmetic (TensorFlow "SAME"/"VALID")
# --------------------------------------------------------------------------------------

def same_pad(size: int, k: int, stride: int, dil: int = 1) -> Tuple[int, int, int]:
    """TF SAME rule: out = ceil(in/s); total = max((out-1)*s + k_eff - in, 0);
    before = total // 2, after = total - before.  Returns (out, before, after)."""
    k_eff = (k - 1) * dil + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    before = total // 2
    return out, before, total - before


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------------------
# convolutions
# --------------------------------------------------------------------------------------

def conv2d(x, w, b=None, stride=1, dil=1, padding="same"):
    """Keras Conv2D (cross-correlation, kernel HWIO, NHWC), e.g. src/models.py:1056-1060,
    1090-1096, 1105-1133.  `padding` is 'same' (TF asymmetric rule) or 'valid'."""
    kh, kw = int(w.shape[0]), int(w.shape[1])
    xn = _nchw(x)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], kh, stride, dil)
        _, pl, pr = same_pad(x.shape[2], kw, stride, dil)
        xn = F.pad(xn, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError(padding)
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride, dilation=dil)
    return _nhwc(y)


def conv2d_transpose(x, w, b=None, stride=2):
    """Keras Conv2DTranspose(padding='same') -> tf.nn.conv2d_transpose with output = in*stride
    (src/models.py:1244-1250, 1311-1317).  Kernel layout (kh, kw, out, in).  It is the gradient of
    the SAME forward conv on the (in*stride)-sized map, so the full transposed result is cropped
    starting at that conv's pad_before."""
    kh, kw = int(w.shape[0]), int(w.shape[1])
    H, W = x.shape[1], x.shape[2]
    _, pt, _ = same_pad(H * stride, kh, stride)
    _, pl, _ = same_pad(W * stride, kw, stride)
    full = F.conv_transpose2d(_nchw(x), w.permute(3, 2, 0, 1), None, stride=stride)
    # full size is (H-1)*s + k which can be smaller than pad + H*s (k < s never happens here)
    need_h, need_w = pt + H * stride, pl + W * stride
    if full.shape[2] < need_h or full.shape[3] < need_w:
        full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    y = full[:, :, pt:pt + H * stride, pl:pl + W * stride]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


# --------------------------------------------------------------------------------------
# normalisation / activations / dropout
# --------------------------------------------------------------------------------------

BN_EPS = 1e-3          # Keras BatchNormalization default epsilon
BN_MOMENTUM = 0.1      # src/models.py:1064,1100,... (weight of the OLD moving value)


def batch_norm_train(x, gamma, beta, eps=BN_EPS, stats_of=None):
    """Keras BatchNormalization(axis=-1) in the training phase: statistics over N,H,W, biased
    variance.  Returns (y, batch_mean, batch_var).  `stats_of` (default x itself): the tensor the statistics are taken
    from, for evaluations that model a stored, rounded x whose statistics came from the unrounded values."""
    s = x if stats_of is None else stats_of
    mean = s.mean(dim=(0, 1, 2))
    var = ((s - mean) ** 2).mean(dim=(0, 1, 2))
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, mean, var


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    return (x - moving_mean) * torch.rsqrt(moving_var + eps) * gamma + beta


def bn_moving_update(moving, batch, momentum=BN_MOMENTUM):
    """Keras K.moving_average_update: moving <- momentum*moving + (1-momentum)*batch."""
    return momentum * moving + (1.0 - momentum) * batch


def prelu(x, alpha):
    """Keras PReLU(shared_axes=[1,2]): relu(x) - alpha*relu(-x), alpha shape (C,)."""
    return torch.relu(x) - alpha * torch.relu(-x)


def prelu_forced(x, alpha, sign):
    """PReLU with the branch of every element given instead of decided by x: sign = +1 takes x, -1 takes alpha * x, 0 the
    kink (output and gradient 0).  Used by parity tests that replay the branches another evaluation took, so that two
    correct evaluations of a piecewise-linear network compare like evaluations of a smooth one."""
    pos = (sign > 0).to(x.dtype)
    neg = (sign < 0).to(x.dtype)
    return x * (pos + alpha * neg)


def max_pool_forced(x, idx):
    """2x2 pooling with the selection given (per-image flat NHWC indices as max_pool_with_argmax returns them)."""
    B = x.shape[0]
    return torch.gather(x.reshape(B, -1), 1, idx.reshape(B, -1)).reshape(idx.shape)


class _RoundBf16(torch.autograd.Function):
    """Identity whose value (forward) and / or incoming gradient (backward) is rounded to bfloat16, round to nearest even:
    the effect of storing a tensor and / or its gradient in bfloat16 between two float32 computations."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


def round_bf16(x, fwd=True, bwd=True):
    return _RoundBf16.apply(x, fwd, bwd)


def leaky_relu(x, slope=0.1):
    return torch.relu(x) - slope * torch.relu(-x)


def spatial_dropout(x, keep_scale):
    """Keras SpatialDropout2D with noise shape (B,1,1,C); `keep_scale` [B,C] already holds
    0 or 1/(1-rate).  Explicit mask so that both sides of a parity test see the same draw."""
    return x * keep_scale[:, None, None, :]


# --------------------------------------------------------------------------------------
# pooling family
# --------------------------------------------------------------------------------------

def max_pool_2x2(x):
    """Keras MaxPooling2D() default: 2x2 stride 2 'valid' (src/models.py:1062,1159)."""
    return _nhwc(F.max_pool2d(_nchw(x), 2, 2))


def max_pool_with_argmax_np(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """tf.nn.max_pool_with_argmax(ksize 2x2, stride 2, SAME) as used by
    src/layers/pooling.py:43-49.  Window scanned h then w, strict '>' update so the first maximum
    wins; index = (h*W + w)*C + c inside the image (batch not included); windows are clipped at
    the border (SAME padding never wins).  Returns (pooled, int64 indices).  Pure numpy loop over
    the 4 window taps - integer results are exact."""
    B, H, W, C = x.shape
    Ho, Wo = -(-H // 2), -(-W // 2)
    best = np.full((B, Ho, Wo, C), -np.inf, dtype=x.dtype)
    idx = np.zeros((B, Ho, Wo, C), dtype=np.int64)
    c_ar = np.arange(C, dtype=np.int64)[None, None, None, :]
    for dh in range(2):
        for dw in range(2):
            hs = np.arange(Ho) * 2 + dh
            ws = np.arange(Wo) * 2 + dw
            vh = hs < H
            vw = ws < W
            hs_c = np.minimum(hs, H - 1)
            ws_c = np.minimum(ws, W - 1)
            cand = x[:, hs_c][:, :, ws_c]
            valid = (vh[:, None] & vw[None, :])[None, :, :, None]
            flat = ((hs_c[:, None] * W + ws_c[None, :]).astype(np.int64))[None, :, :, None] * C + c_ar
            first = (dh == 0 and dw == 0)
            take = valid & ((cand > best) | first)
            best = np.where(take, cand, best)
            idx = np.where(take, flat, idx)
    return best, idx


def max_unpool_np(vals: np.ndarray, idx: np.ndarray, size=(2, 2)) -> np.ndarray:
    """src/layers/pooling.py:71-103: scatter_nd of the values into a zero [B, 2H*2W*C] buffer at
    the per-image flat indices, reshaped to [B,2H,2W,C].  (Duplicate indices would sum in
    tf.scatter_nd; 2x2/2 pooling indices are unique.)"""
    B, H, W, C = vals.shape
    out = np.zeros((B, H * size[0] * W * size[1] * C), dtype=vals.dtype)
    b = np.arange(B)[:, None]
    np.add.at(out, (np.broadcast_to(b, (B, H * W * C)), idx.reshape(B, -1).astype(np.int64)), vals.reshape(B, -1))
    return out.reshape(B, H * size[0], W * size[1], C)


def max_pool_with_argmax(x: torch.Tensor):
    """Differentiable twin (gradient flows to the argmax position only)."""
    B, H, W, C = x.shape
    pooled_np, idx_np = max_pool_with_argmax_np(x.detach().numpy())
    idx = torch.from_numpy(idx_np)
    pooled = torch.gather(x.reshape(B, -1), 1, idx.reshape(B, -1)).reshape(idx.shape)
    return pooled, idx


def max_unpool(vals: torch.Tensor, idx: torch.Tensor, size=(2, 2)):
    B, H, W, C = vals.shape
    out = torch.zeros((B, H * size[0] * W * size[1] * C), dtype=vals.dtype)
    out = out.scatter_add(1, idx.reshape(B, -1), vals.reshape(B, -1))
    return out.reshape(B, H * size[0], W * size[1], C)


def upsample_nn2(x):
    """Keras UpSampling2D((2,2)): nearest repeat (src/models.py:1287)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def channel_zero_pad(x, c_out):
    """Permute + ZeroPadding2D + Permute of src/models.py:1160-1166: zeros appended at the high
    end of the channel axis."""
    return F.pad(x, (0, c_out - x.shape[3]))


# --------------------------------------------------------------------------------------
# loss (src/losses.py:243-265, 310-331, 341-374)
# --------------------------------------------------------------------------------------

def sparse_weighted_pixelwise_xent(logits, labels, weights):
    """reduce_mean over ALL B*H*W pixels of (lse(z) - z[y]) * w  (not divided by sum(w))."""
    lse = torch.logsumexp(logits, dim=-1)
    picked = torch.gather(logits, -1, labels.long().unsqueeze(-1)).squeeze(-1)
    return ((lse - picked) * weights).mean()


def segmentation_cce_lambda_loss(logits, labels, weights, num_unlabeled: int):
    """segmentation_categorical_cross_entropy_lambda_loss: weights are sliced to the first
    B - num_unlabeled rows (src/losses.py:327-329); logits/labels are NOT sliced (src/losses.py:369),
    so TF would raise a shape error for num_unlabeled > 0; the supervised trainer always passes 0."""
    B = labels.shape[0]
    num_labeled = B - int(num_unlabeled)
    if num_labeled != B:
        raise ValueError("supervised lambda loss requires num_unlabeled == 0 (shape mismatch in the reference)")
    return sparse_weighted_pixelwise_xent(logits, labels, weights[0:num_labeled])


def mean_teacher_consistency_cost(y_pred, mt_pred, consistency_coefficient):
    """src/losses.py:195-215: coefficient * mean over pixels of the class-mean squared difference of the softmax
    outputs.  The teacher logits are an input (no gradient flows into them)."""
    s = torch.softmax(y_pred, dim=-1)
    t = torch.softmax(mt_pred.detach(), dim=-1)
    mse = torch.mean((t - s) ** 2, dim=-1)
    return consistency_coefficient * torch.mean(mse)


def segmentation_mean_teacher_lambda_loss(logits, labels, weights, num_unlabeled: int, mt_pred, consistency_coefficient):
    """src/losses.py:377-436: classification cost over the first B - num_unlabeled rows (mean over THEIR pixels,
    src/losses.py:416-422) plus the consistency cost over every row (src/losses.py:428)."""
    B = labels.shape[0]
    num_labeled = B - int(num_unlabeled)
    cls = sparse_weighted_pixelwise_xent(logits[0:num_labeled], labels[0:num_labeled], weights[0:num_labeled])
    return cls + mean_teacher_consistency_cost(logits, mt_pred, consistency_coefficient)


SUPERPIXEL_EPSILON = 10e-7     # src/settings.py:6


def unlabeled_superpixel_cost(y_true_unlabeled, y_pred_unlabeled, coefficient, epsilon=SUPERPIXEL_EPSILON):
    """src/losses.py:108-192: Sobel responses of the softmax maps (depthwise cross-correlation, SAME zero padding),
    clipped from below at epsilon (tf.clip_by_value passes the gradient inside [min, max] only), vector length over the
    classes of the per-class magnitudes, masked by clip(y_true, 0, 1), mean over batch and space, times the coefficient."""
    dt = y_pred_unlabeled.dtype
    m = torch.clamp(y_true_unlabeled, 0, 1).to(dt)
    s = torch.softmax(y_pred_unlabeled, dim=-1)
    C = s.shape[-1]
    sx = torch.tensor([[1.0, 2.0, 1.0], [0.0, 0.0, 0.0], [-1.0, -2.0, -1.0]], dtype=dt)
    sy = sx.t()
    x = _nchw(s)
    gx = _nhwc(F.conv2d(x, sx.reshape(1, 1, 3, 3).repeat(C, 1, 1, 1), padding=1, groups=C))
    gy = _nhwc(F.conv2d(x, sy.reshape(1, 1, 3, 3).repeat(C, 1, 1, 1), padding=1, groups=C))
    big = torch.finfo(torch.float32).max
    gx = torch.clamp(gx, min=epsilon, max=big)
    gy = torch.clamp(gy, min=epsilon, max=big)
    g_mag = torch.sqrt(gx ** 2 + gy ** 2)
    g_dot = torch.linalg.vector_norm(g_mag, dim=-1)
    return coefficient * torch.mean(g_dot * m)


def segmentation_superpixel_lambda_loss(logits, labels, weights, num_unlabeled: int, sp_coefficient):
    """src/losses.py:439-492: classification cost over the labeled rows + superpixel cost over the unlabeled rows
    (0 when the batch has none)."""
    B = labels.shape[0]
    nl = B - int(num_unlabeled)
    loss = sparse_weighted_pixelwise_xent(logits[0:nl], labels[0:nl], weights[0:nl])
    if num_unlabeled > 0:
        loss = loss + unlabeled_superpixel_cost(labels[nl:], logits[nl:], sp_coefficient)
    return loss


def segmentation_mean_teacher_superpixel_lambda_loss(logits, labels, weights, num_unlabeled: int, mt_pred, mt_coefficient,
                                                     sp_coefficient):
    """src/losses.py:495-572."""
    B = labels.shape[0]
    nl = B - int(num_unlabeled)
    loss = sparse_weighted_pixelwise_xent(logits[0:nl], labels[0:nl], weights[0:nl])
    loss = loss + mean_teacher_consistency_cost(logits, mt_pred, mt_coefficient)
    if num_unlabeled > 0:
        loss = loss + unlabeled_superpixel_cost(labels[nl:], logits[nl:], sp_coefficient)
    return loss


def classification_weighted_categorical_crossentropy(y_true, y_pred, class_weights, epsilon=SUPERPIXEL_EPSILON):
    """src/losses.py:579-588: -mean_n sum_c y_true * log(clip(softmax(y_pred), eps, 1)) * class_weights (tf.clip_by_value
    passes the gradient inside [eps, 1] only, like torch.clamp)."""
    sm = torch.clamp(torch.softmax(y_pred, dim=-1), epsilon, 1.0)
    return -torch.mean(torch.sum(y_true * torch.log(sm) * class_weights, dim=-1))


def ema_update_np(teacher, student, a):
    """src/trainers.py:1692-1706 on lists of float32 numpy arrays; `a` is a Python float, so numpy keeps float32."""
    return [a * t + ((1.0 - a) * s) for t, s in zip(teacher, student)]


def class_weight_map(labels, class_weights):
    """src/losses.py:270-298 (while_loop over classes) == class_weights[labels];
    src/generators.py:1814-1825 builds the same map on the host."""
    return class_weights[labels.long()]


# --------------------------------------------------------------------------------------
# metrics (src/metrics.py:250-273, 89-130; src/scripts/calculate_metrics_from_cfms.py:12-53)
# --------------------------------------------------------------------------------------

def confusion_matrix_np(labels: np.ndarray, logits: np.ndarray, num_classes: int,
                        ignore_classes: Sequence[int] = ()) -> np.ndarray:
    """cm[label, pred] += 1 for every pixel whose label is not ignored; pred = first-index argmax.
    int64, rows = truth (tf.confusion_matrix)."""
    lab = labels.reshape(-1).astype(np.int64)
    pred = np.argmax(logits.reshape(-1, logits.shape[-1]), axis=-1).astype(np.int64)  # first max wins
    keep = np.ones_like(lab, dtype=bool)
    for c in ignore_classes:
        keep &= lab != c
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    np.add.at(cm, (lab[keep], pred[keep]), 1)
    return cm


def cfm_accuracy(cm: np.ndarray) -> float:
    """calculate_metrics_from_cfms.py:12-15 (== tf.metrics.accuracy with the ignore weights,
    because ignored rows never enter the matrix)."""
    tot = cm.sum()
    return float(np.trace(cm)) / float(tot) if tot else 0.0


def cfm_mpca(cm: np.ndarray, ignored: Sequence[int] = ()) -> float:
    """calculate_metrics_from_cfms.py:18-32."""
    per = cm.sum(axis=1).astype(np.float64)
    per[per == 0] = 1
    acc = list(np.diagonal(cm).astype(np.float32).astype(np.float64) / per)  # float32 numerator as in the script
    for c in sorted(ignored, reverse=True):
        del acc[c]
    return float(np.mean(acc))


def cfm_miou(cm: np.ndarray, ignored: Sequence[int] = ()) -> float:
    """calculate_metrics_from_cfms.py:35-53."""
    tp = np.diagonal(cm).astype(np.float32).astype(np.float64)  # float32 numerator as in the script
    den = cm.sum(axis=0) + cm.sum(axis=1) - np.diagonal(cm)
    den = den.astype(np.float64)
    den[den == 0] = 1
    iou = list(tp / den)
    for c in sorted(ignored, reverse=True):
        del iou[c]
    return float(np.mean(iou))


def tf_mean_per_class_accuracy(cm: np.ndarray) -> float:
    """tf.metrics.mean_per_class_accuracy: mean over ALL classes of diag/row, 0 where the row is
    empty (so an ignored class contributes a 0 term)."""
    rows = cm.sum(axis=1).astype(np.float64)
    diag = np.diagonal(cm).astype(np.float64)
    per = np.where(rows > 0, diag / np.where(rows > 0, rows, 1), 0.0)
    return float(per.mean())


def tf_mean_iou(cm: np.ndarray, only_present: bool = True) -> float:
    """tf.metrics.mean_iou: IoU_c = cm_cc / (row+col-cm_cc), zero denominators -> 1;
    TF >= 1.5 averages over classes with non-zero denominator, TF <= 1.4 over all classes."""
    tp = np.diagonal(cm).astype(np.float64)
    den = (cm.sum(axis=0) + cm.sum(axis=1) - np.diagonal(cm)).astype(np.float64)
    iou = tp / np.where(den > 0, den, 1)
    if only_present:
        n = (den > 0).sum()
        return float(iou.sum() / n) if n else 0.0
    return float(iou.mean())


# --------------------------------------------------------------------------------------
# optimizers (src/keras_extensions/extended_optimizers.py:35-67, 117-162)
# --------------------------------------------------------------------------------------

def adam_step(params, grads, ms, vs, iterations: int, lr=1e-3, beta_1=0.9, beta_2=0.999, eps=1e-8,
              decay=0.0, lr_scalers=None, vhats=None):
    """One Adam.get_updates application.  `iterations` is the value BEFORE the update (t = it+1).
    Lists are updated in place (torch tensors).  AMSGrad when vhats is given."""
    lr_scalers = lr_scalers or {}
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * iterations))
    t = iterations + 1
    lr_t0 = lr * (math.sqrt(1.0 - beta_2 ** t) / (1.0 - beta_1 ** t))
    for i, (p, g) in enumerate(zip(params, grads)):
        lr_t = lr_t0 * lr_scalers.get(i, 1.0)
        ms[i] = beta_1 * ms[i] + (1.0 - beta_1) * g
        vs[i] = beta_2 * vs[i] + (1.0 - beta_2) * g * g
        if vhats is not None:
            vhats[i] = torch.maximum(vhats[i], vs[i])
            params[i] = p - lr_t * ms[i] / (torch.sqrt(vhats[i]) + eps)
        else:
            params[i] = p - lr_t * ms[i] / (torch.sqrt(vs[i]) + eps)
    return params


def sgd_step(params, grads, moments, iterations: int, lr=0.01, momentum=0.0, decay=0.0, nesterov=False,
             lr_scalers=None):
    lr_scalers = lr_scalers or {}
    if decay > 0:
        lr = lr * (1.0 / (1.0 + decay * iterations))
    for i, (p, g) in enumerate(zip(params, grads)):
        lr_i = lr * lr_scalers.get(i, 1.0)
        v = momentum * moments[i] - lr_i * g
        moments[i] = v
        params[i] = p + momentum * v - lr_i * g if nesterov else p + v
    return params


def np_softmax(x: np.ndarray, axis=-1) -> np.ndarray:
    """src/predict.py:37-79 (stable softmax)."""
    z = x - np.max(x, axis=axis, keepdims=True)
    e = np.exp(z)
    return e / np.sum(e, axis=axis, keepdims=True)


def mfb_class_weights(class_probabilities) -> np.ndarray:
    """src/utils/dataset_utils.py:498-513."""
    p = np.asarray(class_probabilities, dtype=np.float64)
    med = np.median(p[p > 0])
    return np.where(p > 0, med / np.where(p > 0, p, 1.0), 0.0)


def enet_class_weights(class_probabilities, c=1.02) -> np.ndarray:
    """src/utils/dataset_utils.py:515-529."""
    p = np.asarray(class_probabilities, dtype=np.float64)
    return np.where(p > 0, 1.0 / np.log(c + p), 0.0)
