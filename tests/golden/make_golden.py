"""Generates the golden vectors under tests/golden/ from the fp64 CPU oracle.

    python tests/golden/make_golden.py

The reference (TensorFlow/Keras, Python 2) cannot be imported or run in the build container, and it ships no test
vectors for this path, so these fixtures pin the ORACLE (they guard it against drift and give the GPU tests a fixed
target that does not depend on re-running the oracle); they are not outputs of the reference itself.

Files (all float64 unless noted; inputs are regenerated from the recorded seeds where they are large):
  ops.npz        per-operator cases: SAME stride-2 convolution, dilated convolution, transposed convolution,
                 BN training statistics, max-pool with argmax (with ties), unpooling, weighted cross entropy,
                 confusion matrix, metrics, three Adam steps, three SGD(momentum, nesterov) steps, the mean-teacher
                 lambda loss with one unlabeled row, the teacher's EMA update, the superpixel lambda loss with two unlabeled rows
  step_<model>.npz  one supervised training step of every model at a tiny shape: loss, logits, dlogits,
                 confusion matrix, per-tensor gradient checksums (sum, sum of squares) and a few full gradients
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import enet_oracle as E          # noqa: E402
from oracle import tf_keras_ops as K         # noqa: E402

STEP_CASES = {
    # model: (batch, height, width, seed)
    "enet-naive-upsampling": (2, 32, 48, 21),
    "enet-max-unpooling": (2, 48, 32, 22),
    "enet-naive-upsampling-enhanced": (2, 32, 32, 23),
    "unet": (1, 32, 32, 24),
    "segnet": (1, 32, 32, 25),
}
FULL_GRADS = {   # gradient tensors stored in full (first layer, one mid layer, the head)
    "enet": ["initial_block_conv2d/kernel", "en_bn_2.3_aconv2d_2/kernel", "en_bn_3.8_prelu_3/alpha", "logits/kernel", "logits/bias"],
    "vgg": ["encoder_block1_conv1/kernel", "encoder_block1_conv1_normalization/gamma", "decoder_block4_conv2_normalization/beta",
            "logits/kernel", "logits/bias"],
}


def t(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def make_ops():
    rng = np.random.RandomState(1234)
    out = {}
    # convolution, TF SAME, stride 2, odd extent (pad_before 0 / pad_after 1 along an even side ...)
    x = rng.standard_normal((2, 7, 10, 5))
    w = rng.standard_normal((3, 3, 5, 6)) * 0.3
    b = rng.standard_normal(6)
    out["conv_s2_x"], out["conv_s2_w"], out["conv_s2_b"] = x, w, b
    out["conv_s2_y"] = K.conv2d(t(x), t(w), t(b), stride=2, dil=1, padding="same").numpy()
    # dilated 3x3 (dilation larger than half the map)
    x = rng.standard_normal((1, 9, 9, 4))
    w = rng.standard_normal((3, 3, 4, 4)) * 0.3
    out["conv_d4_x"], out["conv_d4_w"] = x, w
    out["conv_d4_y"] = K.conv2d(t(x), t(w), None, stride=1, dil=4, padding="same").numpy()
    # asymmetric 1x5
    x = rng.standard_normal((1, 6, 11, 4))
    w = rng.standard_normal((1, 5, 4, 8)) * 0.3
    out["conv_1x5_x"], out["conv_1x5_w"] = x, w
    out["conv_1x5_y"] = K.conv2d(t(x), t(w), None, padding="same").numpy()
    # transposed 3x3 stride 2 and 2x2 stride 2 (Keras kernel layout kh, kw, out, in)
    x = rng.standard_normal((2, 4, 5, 4))
    w = rng.standard_normal((3, 3, 6, 4)) * 0.3
    b = rng.standard_normal(6)
    out["tconv3_x"], out["tconv3_w"], out["tconv3_b"] = x, w, b
    out["tconv3_y"] = K.conv2d_transpose(t(x), t(w), t(b), stride=2).numpy()
    w = rng.standard_normal((2, 2, 5, 4)) * 0.3
    b = rng.standard_normal(5)
    out["tconv2_w"], out["tconv2_b"] = w, b
    out["tconv2_y"] = K.conv2d_transpose(t(x), t(w), t(b), stride=2).numpy()
    # BN training statistics + moving update + PReLU
    x = rng.standard_normal((2, 5, 6, 8)) * 2 + 1
    g, be = rng.uniform(0.5, 1.5, 8), rng.standard_normal(8)
    y, mean, var = K.batch_norm_train(t(x), t(g), t(be))
    out["bn_x"], out["bn_gamma"], out["bn_beta"] = x, g, be
    out["bn_y"], out["bn_mean"], out["bn_var"] = y.numpy(), mean.numpy(), var.numpy()
    out["bn_moving_var_next"] = K.bn_moving_update(t(np.ones(8)), var).numpy()
    al = rng.uniform(0.05, 0.3, 8)
    out["prelu_alpha"] = al
    out["prelu_y"] = K.prelu(y, t(al)).numpy()
    # max pool with argmax incl. exact ties (integers) and unpool
    x = rng.randint(0, 3, size=(2, 6, 8, 4)).astype(np.float64)
    p, idx = K.max_pool_with_argmax_np(x)
    out["pool_x"], out["pool_y"], out["pool_idx"] = x, p, idx.astype(np.int64)
    out["unpool_y"] = K.max_unpool_np(p, idx)
    # weighted pixelwise cross entropy, gradient, confusion matrix, metrics
    logits = rng.standard_normal((2, 8, 8, 24)) * 3
    labels = rng.randint(0, 24, size=(2, 8, 8))
    cw = np.concatenate([[0.0], 0.5 + np.arange(1, 24) / 23.0])
    wts = cw[labels]
    lg = t(logits).requires_grad_(True)
    loss = K.segmentation_cce_lambda_loss(lg, torch.tensor(labels), t(wts), 0)
    (dl,) = torch.autograd.grad(loss, [lg])
    cm = K.confusion_matrix_np(labels, logits, 24, [0])
    out["xent_logits"], out["xent_labels"], out["xent_weights"] = logits, labels.astype(np.int64), wts
    out["xent_loss"], out["xent_dlogits"], out["xent_cm"] = np.float64(loss.item()), dl.numpy(), cm.astype(np.int64)
    out["metrics"] = np.array([K.cfm_accuracy(cm), K.tf_mean_per_class_accuracy(cm), K.tf_mean_iou(cm, True),
                               K.tf_mean_iou(cm, False), K.cfm_mpca(cm, [0]), K.cfm_miou(cm, [0])])
    # optimizers
    p0 = [rng.standard_normal(17), rng.standard_normal((3, 5))]
    gs = [[rng.standard_normal(17), rng.standard_normal((3, 5))] for _ in range(3)]
    ps = [t(p) for p in p0]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for it in range(3):
        K.adam_step(ps, [t(g) for g in gs[it]], ms, vs, it, lr=5e-4, beta_1=0.9, beta_2=0.999, eps=1e-8,
                    lr_scalers={1: 0.5})
    out["opt_p0_a"], out["opt_p0_b"] = p0
    for it in range(3):
        out["opt_g{}_a".format(it)], out["opt_g{}_b".format(it)] = gs[it]
    out["adam_p3_a"], out["adam_p3_b"] = ps[0].numpy(), ps[1].numpy()
    ps = [t(p) for p in p0]
    mo = [torch.zeros_like(p) for p in ps]
    for it in range(3):
        K.sgd_step(ps, [t(g) for g in gs[it]], mo, it, lr=0.01, momentum=0.9, decay=1e-3, nesterov=True,
                   lr_scalers={1: 0.5})
    out["sgd_p3_a"], out["sgd_p3_b"] = ps[0].numpy(), ps[1].numpy()
    # mean-teacher lambda loss: 3 rows, the last one unlabeled; EMA of the teacher weights
    logits = rng.standard_normal((3, 4, 4, 24)) * 2
    teacher = rng.standard_normal((3, 4, 4, 24)) * 2
    labels = rng.randint(0, 24, size=(3, 4, 4))
    wts = cw[labels]
    lg = t(logits).requires_grad_(True)
    loss = K.segmentation_mean_teacher_lambda_loss(lg, torch.tensor(labels), t(wts), 1, t(teacher), 3.5)
    (dl,) = torch.autograd.grad(loss, [lg])
    out["mt_logits"], out["mt_teacher"], out["mt_labels"], out["mt_weights"] = logits, teacher, labels.astype(np.int64), wts
    out["mt_loss"], out["mt_dlogits"] = np.float64(loss.item()), dl.numpy()
    out["mt_cost"] = np.float64(K.mean_teacher_consistency_cost(t(logits), t(teacher), 3.5).item())
    tw = [rng.standard_normal(11).astype(np.float32), rng.standard_normal((2, 3)).astype(np.float32)]
    sw = [rng.standard_normal(11).astype(np.float32), rng.standard_normal((2, 3)).astype(np.float32)]
    ew = K.ema_update_np(tw, sw, 0.99)
    out["ema_t_a"], out["ema_t_b"], out["ema_s_a"], out["ema_s_b"] = tw[0], tw[1], sw[0], sw[1]
    out["ema_out_a"], out["ema_out_b"] = ew[0], ew[1]
    # superpixel lambda loss: 4 rows of 6x5 pixels, the last two unlabeled with superpixel-border masks (0 = border)
    logits = rng.standard_normal((4, 6, 5, 24)) * 2
    labels = rng.randint(0, 24, size=(4, 6, 5))
    labels[2:] = rng.randint(0, 4, size=(2, 6, 5))          # superpixel ids; 0 marks a border
    wts = cw[labels]
    lg = t(logits).requires_grad_(True)
    loss = K.segmentation_superpixel_lambda_loss(lg, torch.tensor(labels), t(wts), 2, 1.75)
    (dl,) = torch.autograd.grad(loss, [lg])
    out["sp_logits"], out["sp_labels"], out["sp_weights"] = logits, labels.astype(np.int64), wts
    out["sp_loss"], out["sp_dlogits"] = np.float64(loss.item()), dl.numpy()
    out["sp_cost"] = np.float64(K.unlabeled_superpixel_cost(torch.tensor(labels[2:]), t(logits[2:]), 1.75).item())
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def step_inputs(model):
    """Inputs of a step case, regenerated from seeds (shared with the tests)."""
    B, H, W, seed = STEP_CASES[model]
    nc = 24
    w0 = E.init_weights(model, nc, seed=seed, perturb=True)
    cw = np.ones(nc, dtype=np.float32)
    cw[0] = 0
    cw[1:] = 0.5 + np.arange(1, nc) / 23.0
    images, labels, pw = E.synthetic_batch(B, H, W, nc, seed=seed + 100, block=8, class_weights=cw)
    masks = E.make_dropout_masks(model, B, seed + 7)
    return w0, images, labels, pw, masks


def make_step(model):
    w0, images, labels, pw, masks = step_inputs(model)
    r = E.train_step(model, w0, images, labels, pw, masks, collect=True)
    out = {"loss": np.float64(r["loss"]), "logits": r["logits"].astype(np.float32), "cm": r["cm"].astype(np.int64),
           "dlogits": r["dlogits"].astype(np.float32)}
    keys = list(r["grads"].keys())
    out["grad_keys"] = np.array(keys)
    out["grad_sum"] = np.array([r["grads"][k].sum() for k in keys])
    out["grad_sumsq"] = np.array([(r["grads"][k] ** 2).sum() for k in keys])
    out["grad_absmax"] = np.array([np.abs(r["grads"][k]).max() for k in keys])
    for k in FULL_GRADS["vgg" if model in E.VGG_STYLE else "enet"]:
        out["grad/" + k] = r["grads"][k]
    bn_keys = [k for k in r["new_weights"] if k.endswith("moving_variance")]
    out["moving_variance_keys"] = np.array(bn_keys)
    out["moving_variance_mean"] = np.array([r["new_weights"][k].mean() for k in bn_keys])
    np.savez_compressed(os.path.join(HERE, "step_{}.npz".format(model)), **out)
    print(model, "loss", r["loss"])


if __name__ == "__main__":
    make_ops()
    if "--ops-only" not in sys.argv:
        for m in STEP_CASES:
            make_step(m)
