"""CPU oracle: the U-Net and SegNet-Basic builders of the reference's src/models.py restated on top of
oracle/tf_keras_ops.py (PyTorch CPU, NHWC, autograd for the backward pass).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
PARITY UNPINNED: the reference holds no test vectors for this path; see tf_keras_ops.py.

Topologies follow:
  unet   : src/models.py:789-825   (UNetModel._build_model)
  segnet : src/models.py:974-1012  (SegNetBasicModel._build_model)
Both are assembled from UNetModel.get_convolution_block / get_encoder_block / get_decoder_block
(src/models.py:828-943):
  convolution block = ZeroPadding2D(1) -> Conv2D(3x3, valid) -> BatchNormalization(momentum 0.1)
                      -> [LeakyReLU(0.1)] -> SpatialDropout2D(0.1)
  encoder block     = n convolution blocks (bias, activation) -> MaxPooling2D(2x2); returns (pre-pool, pooled)
  decoder block     = UpSampling2D(2) -> concatenate([up, skip]) -> n convolution blocks (NO bias, NO activation)
  head              = Conv2D(num_classes, 1x1) named `logits`
The reference leaves its SpatialDropout2D layers unnamed (Keras numbers them); the tables are keyed here by the
convolution block they follow: "{block}_dropout".
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Tuple

import torch

from . import tf_keras_ops as K

DROP_RATE = 0.1      # src/models.py:841
LEAKY_ALPHA = 0.1    # src/models.py:840


def topology(model_name: str):
    """(encoder blocks, connecting widths, decoder blocks): encoder (prefix, n_conv, filters);
    decoder (prefix, n_conv, filters, index of the encoder block whose pre-pool output is concatenated)."""
    if model_name == "unet":
        enc = [("encoder_block1", 2, 64), ("encoder_block2", 2, 128), ("encoder_block3", 2, 256), ("encoder_block4", 2, 512)]
        connect = [("connecting_block_conv1", 1024), ("connecting_block_conv2", 1024)]
        dec = [("decoder_block1", 2, 512, 3), ("decoder_block2", 2, 256, 2), ("decoder_block3", 2, 128, 1),
               ("decoder_block4", 2, 64, 0)]
    elif model_name == "segnet":
        enc = [("encoder_block1", 2, 64), ("encoder_block2", 2, 128), ("encoder_block3", 3, 256), ("encoder_block4", 3, 512)]
        connect = []
        dec = [("decoder_block2", 3, 512, 3), ("decoder_block3", 3, 256, 2), ("decoder_block4", 2, 128, 1),
               ("decoder_block5", 2, 64, 0)]
    else:
        raise ValueError("Unknown model name: {}".format(model_name))
    return enc, connect, dec


def conv_blocks(model_name: str, encoder_only: bool = False) -> List[Tuple[str, int, int, bool, bool]]:
    """Every convolution block in creation order: (name, cin, cout, use_bias, use_activation)."""
    enc, connect, dec = topology(model_name)
    if encoder_only:      # src/models.py:985-993: the head sits on pool4
        connect, dec = [], []
    out = []
    c = 3
    widths = []
    for prefix, n, f in enc:
        for i in range(n):
            out.append(("{}_conv{}".format(prefix, i + 1), c, f, True, True))
            c = f
        widths.append(f)
    for name, f in connect:
        out.append((name, c, f, True, True))
        c = f
    for prefix, n, f, skip in dec:
        c = c + widths[skip]
        for i in range(n):
            out.append(("{}_conv{}".format(prefix, i + 1), c, f, False, False))
            c = f
    return out


def weight_spec(model_name: str, num_classes: int, encoder_only: bool = False):
    """[(key, shape, initializer, fan_in)] in layer-creation order (Keras get_weights() order)."""
    spec = []
    last = 3
    for name, cin, cout, bias, _ in conv_blocks(model_name, encoder_only):
        spec.append((name + "/kernel", (3, 3, cin, cout), "he_normal", 9 * cin))
        if bias:
            spec.append((name + "/bias", (cout,), "zeros", 0))
        bn = name + "_normalization"
        spec.append((bn + "/gamma", (cout,), "ones", 0))
        spec.append((bn + "/beta", (cout,), "zeros", 0))
        spec.append((bn + "/moving_mean", (cout,), "zeros", 0))
        spec.append((bn + "/moving_variance", (cout,), "ones", 0))
        last = cout
    if encoder_only:      # Dense: kernel (in, out), glorot_uniform
        spec.append(("logits/kernel", (last, num_classes), "glorot_uniform", 0))
        spec.append(("logits/bias", (num_classes,), "zeros", 0))
        return spec
    spec.append(("logits/kernel", (1, 1, last, num_classes), "he_normal", last))
    spec.append(("logits/bias", (num_classes,), "zeros", 0))
    return spec


def dropout_layer_rates(model_name: str, encoder_only: bool = False) -> "OrderedDict[str, Tuple[float, int]]":
    return OrderedDict((name + "_dropout", (DROP_RATE, cout)) for name, _, cout, _, _ in conv_blocks(model_name, encoder_only))


def _conv_block(c, x, name, act):
    # ZeroPadding2D((1,1)) + 3x3 'valid' == 3x3 with one zero pixel on every side
    y = c.conv(name, x, padding="same")
    y = c.bn(name + "_normalization", y)
    if act:
        # c.forced[name + "_activation"]: sign per element (+1 / -1 / 0) of the LeakyReLU input as another evaluation saw it;
        # the branch is then taken from there instead of from y (tests that replay the branches a device step took)
        fk = name + "_activation"
        y = c.keep(fk, K.prelu_forced(y, LEAKY_ALPHA, c.forced[fk]) if fk in c.forced else K.leaky_relu(y, LEAKY_ALPHA))
    return c.keep(name + "_dropout", c.sdrop(name + "_dropout", y))


def forward(model_name: str, c, images, encoder_only: bool = False):
    """`c` is the oracle context of enet_oracle (_Ctx): named conv / bn / sdrop application."""
    enc, connect, dec = topology(model_name)
    x = images
    skips = []
    for prefix, n, _ in enc:
        for i in range(n):
            x = _conv_block(c, x, "{}_conv{}".format(prefix, i + 1), True)
        skips.append(x)
        fk = prefix + "_pool/argmax"      # forced selection: per-image flat NHWC indices of the window maxima
        x = c.keep(prefix + "_pool", K.max_pool_forced(x, c.forced[fk]) if fk in c.forced else K.max_pool_2x2(x))
    if encoder_only:      # AveragePooling2D(pool4's spatial size) -> Flatten -> Dense (src/models.py:985-993)
        pooled = c.keep("avg_pool2d", x.mean(dim=(1, 2)))
        return pooled @ c.p["logits/kernel"] + c.p["logits/bias"]
    for name, _ in connect:
        x = _conv_block(c, x, name, True)
    for prefix, n, _, skip in dec:
        x = torch.cat([K.upsample_nn2(x), skips[skip]], dim=3)      # src/models.py:920-927: [up, skip]
        for i in range(n):
            x = _conv_block(c, x, "{}_conv{}".format(prefix, i + 1), False)
    return c.conv("logits", x, padding="valid")
