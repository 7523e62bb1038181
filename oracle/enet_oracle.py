"""CPU oracle: the ENet encoder-decoders of the reference's src/models.py restated on top of
oracle/tf_keras_ops.py (PyTorch CPU, NHWC, autograd for the backward pass).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
PARITY UNPINNED: the reference holds no test vectors for this path; see tf_keras_ops.py.

Weights are a flat dict  "{layer_name}/{kernel|bias|gamma|beta|moving_mean|moving_variance|alpha}"
in Keras layouts (Conv2D kernel HWIO, Conv2DTranspose kernel (kh,kw,out,in), PReLU alpha (1,1,C)).

Topologies follow:
  enet-naive-upsampling : src/models.py:1054-1319
  enet-max-unpooling    : src/models.py:1360-1628
  enet-naive-upsampling-enhanced : src/models.py:1642-1796
Deviation kept on purpose: the naive decoder's second PReLU is named `_prelu_2`
(the reference names both `_prelu_1`, src/models.py:1231,1254, which Keras rejects).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import tf_keras_ops as K
from . import unet_segnet_oracle as V

VGG_STYLE = ("unet", "segnet")   # restated in unet_segnet_oracle.py; weights / step assembly are shared with the ENets

ENC_PATTERN = [("1", {}), ("2", {"dilated": 2}), ("3", {"asymmetric": 5}), ("4", {"dilated": 4}),
               ("5", {}), ("6", {"dilated": 8}), ("7", {"asymmetric": 5}), ("8", {"dilated": 16})]


# --------------------------------------------------------------------------------------
# weight specification, in layer-creation order of the reference builders
# --------------------------------------------------------------------------------------

def _conv(spec, name, kh, kw, cin, cout, bias):
    spec.append((name + "/kernel", (kh, kw, cin, cout), "he_normal", kh * kw * cin))
    if bias:
        spec.append((name + "/bias", (cout,), "zeros", 0))


def _tconv(spec, name, kh, kw, cin, cout):
    # Keras Conv2DTranspose kernel: (kh, kw, out, in); he_normal fan_in = kh*kw*in (as Keras computes
    # it for this layout: receptive field * shape[-2]... kept simple: kh*kw*cin)
    spec.append((name + "/kernel", (kh, kw, cout, cin), "he_normal", kh * kw * cin))
    spec.append((name + "/bias", (cout,), "zeros", 0))


def _bn(spec, name, c):
    spec.append((name + "/gamma", (c,), "ones", 0))
    spec.append((name + "/beta", (c,), "zeros", 0))
    spec.append((name + "/moving_mean", (c,), "zeros", 0))
    spec.append((name + "/moving_variance", (c,), "ones", 0))


def _prelu(spec, name, c):
    spec.append((name + "/alpha", (1, 1, c), "zeros", 0))


def _enc_bottleneck_spec(spec, p, cin, cout, downsample=False, asymmetric=0, dilated=0):
    internal = cout // 4
    s = 2 if downsample else 1
    _conv(spec, p + "_proj_conv2d_1", s, s, cin, internal, False)
    _bn(spec, p + "_bnorm_1", internal)
    _prelu(spec, p + "_prelu_1", internal)
    if asymmetric:
        _conv(spec, p + "_aconv2d_1", 1, asymmetric, internal, internal, False)
        _conv(spec, p + "_aconv2d_2", asymmetric, 1, internal, internal, True)
    elif dilated:
        _conv(spec, p + "_dconv2d", 3, 3, internal, internal, True)
    else:
        _conv(spec, p + "_conv2d_1", 3, 3, internal, internal, True)
    _bn(spec, p + "_bnorm_2", internal)
    _prelu(spec, p + "_prelu_2", internal)
    _conv(spec, p + "_proj_conv2d_2", 1, 1, internal, cout, False)
    _bn(spec, p + "_bnorm_3", cout)
    _prelu(spec, p + "_prelu_3", cout)


def _dec_bottleneck_spec(spec, p, cin, cout, upsample, variant):
    internal = cout // 4
    _conv(spec, p + "_proj_conv2d_1", 1, 1, cin, internal, False)
    _bn(spec, p + "_bnorm_1", internal)
    _prelu(spec, p + "_prelu_1", internal)
    if upsample and variant == "enhanced":
        _conv(spec, p + "_nn_rs_conv_conv2d", 3, 3, internal, internal, True)
    elif upsample:
        _tconv(spec, p + "_tconv2d_1", 3, 3, internal, internal)
    else:
        _conv(spec, p + "_conv2d_1", 3, 3, internal, internal, True)
    _bn(spec, p + "_bnorm_2", internal)
    _prelu(spec, p + "_prelu_2", internal)
    _conv(spec, p + "_proj_conv2d_2", 1, 1, internal, cout, False)
    has_other = (cin != cout) or upsample
    if variant in ("naive", "enhanced"):
        _bn(spec, p + "_bnorm_3", cout)
        if has_other:
            _conv(spec, p + "_other_conv2d", 1, 1, cin, cout, False)
            _bn(spec, p + "_other_bnorm_1", cout)
    else:
        if has_other:
            _conv(spec, p + "_other_conv2d", 1, 1, cin, cout, False)
            _bn(spec, p + "_other_bnorm_1", cout)
        _bn(spec, p + "_bnorm_3", cout)
    _prelu(spec, p + "_prelu_3", cout)


ENCODER_ONLY = "-encoder-only"      # src/models.py:83-124: names of the classification variants


def split_name(model_name: str) -> Tuple[str, bool]:
    """(segmentation model name, encoder_only)."""
    if model_name.endswith(ENCODER_ONLY):
        return model_name[:-len(ENCODER_ONLY)], True
    return model_name, False


def _dense(spec, name, cin, cout):
    """Keras Dense: kernel (in, out) glorot_uniform, bias zeros."""
    spec.append((name + "/kernel", (cin, cout), "glorot_uniform", 0))
    spec.append((name + "/bias", (cout,), "zeros", 0))


def weight_spec(model_name: str, num_classes: int) -> List[Tuple[str, tuple, str, int]]:
    """[(key, shape, initializer, fan_in)] in layer-creation order."""
    model_name, encoder_only = split_name(model_name)
    if model_name in VGG_STYLE:
        return V.weight_spec(model_name, num_classes, encoder_only)
    if model_name == "enet-naive-upsampling":
        variant = "naive"
    elif model_name == "enet-max-unpooling":
        variant = "unpool"
    elif model_name == "enet-naive-upsampling-enhanced":
        variant = "enhanced"
    else:
        raise ValueError("Unknown model name: {}".format(model_name))
    spec: list = []
    spec.append(("initial_block_conv2d/kernel", (3, 3, 3, 13), "glorot_uniform", 0))
    spec.append(("initial_block_conv2d/bias", (13,), "zeros", 0))
    if variant in ("naive", "enhanced"):
        _bn(spec, "initial_block_bnorm", 16)
        _prelu(spec, "initial_block_prelu", 16)
    _enc_bottleneck_spec(spec, "en_bn_1.0", 16, 64, downsample=True)
    for i in range(4):
        _enc_bottleneck_spec(spec, "en_bn_1.{}".format(i + 1), 64, 64)
    _enc_bottleneck_spec(spec, "en_bn_2.0", 64, 128, downsample=True)
    for stage in (2, 3):
        for suffix, kw in ENC_PATTERN:
            _enc_bottleneck_spec(spec, "en_bn_{}.{}".format(stage, suffix), 128, 128, **kw)
    if encoder_only and variant != "enhanced":      # src/models.py:1040-1048, 1346-1354
        _dense(spec, "logits", 128, num_classes)
        return spec
    _dec_bottleneck_spec(spec, "de_bn_4.0", 128, 64, True, variant)
    _dec_bottleneck_spec(spec, "de_bn_4.1", 64, 64, False, variant)
    _dec_bottleneck_spec(spec, "de_bn_4.2", 64, 64, False, variant)
    _dec_bottleneck_spec(spec, "de_bn_5.0", 64, 16, True, variant)
    _dec_bottleneck_spec(spec, "de_bn_5.1", 16, 16, False, variant)
    if variant == "enhanced":
        _conv(spec, "final_nn_rs_conv_conv2d", 3, 3, 16, num_classes, True)
        _bn(spec, "final_nn_rs_conv_bnorm", num_classes)
        _prelu(spec, "final_nn_rs_conv_prelu", num_classes)
        _conv(spec, "decoder_logits" if encoder_only else "logits", 1, 1, num_classes, num_classes, True)
        if encoder_only:      # src/models.py:1665-1668
            _dense(spec, "logits", num_classes, num_classes)
        return spec
    spec.append(("logits/kernel", (2, 2, num_classes, 16), "he_normal", 2 * 2 * 16))
    spec.append(("logits/bias", (num_classes,), "zeros", 0))
    return spec


def init_weights(model_name: str, num_classes: int, seed: int = 14874, perturb: bool = False,
                 dtype=np.float64) -> "OrderedDict[str, np.ndarray]":
    """Synthetic initial weights.  `perturb=True` also randomises BN gamma/beta, PReLU alpha, biases
    and moving statistics so that parity tests exercise every term (Keras defaults are 1/0/0)."""
    rng = np.random.RandomState(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape, init, fan_in in weight_spec(model_name, num_classes):
        if init == "he_normal":
            w = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif init == "glorot_uniform" and len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = rng.uniform(-lim, lim, size=shape)
        elif init == "glorot_uniform":
            kh, kw, cin, cout = shape
            lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
            w = rng.uniform(-lim, lim, size=shape)
        elif init == "ones":
            w = np.ones(shape)
            if perturb:
                if key.endswith("moving_variance"):
                    w = rng.uniform(0.5, 1.5, size=shape)
                else:
                    w = rng.uniform(0.7, 1.3, size=shape)
        else:
            w = np.zeros(shape)
            if perturb:
                if key.endswith("alpha"):
                    w = rng.uniform(0.05, 0.3, size=shape)
                else:
                    w = rng.uniform(-0.2, 0.2, size=shape)
        out[key] = np.ascontiguousarray(w.astype(dtype))
    return out


def dropout_layer_rates(model_name: str) -> "OrderedDict[str, Tuple[float, int]]":
    """{sdrop layer name: (rate, channels)} in creation order."""
    model_name, encoder_only = split_name(model_name)
    if model_name in VGG_STYLE:
        return V.dropout_layer_rates(model_name, encoder_only)
    out: "OrderedDict[str, Tuple[float, int]]" = OrderedDict()
    out["en_bn_1.0_sdrop2d_1"] = (0.01, 64)
    for i in range(4):
        out["en_bn_1.{}_sdrop2d_1".format(i + 1)] = (0.01, 64)
    out["en_bn_2.0_sdrop2d_1"] = (0.1, 128)
    for stage in (2, 3):
        for suffix, _ in ENC_PATTERN:
            out["en_bn_{}.{}_sdrop2d_1".format(stage, suffix)] = (0.1, 128)
    if (model_name == "enet-naive-upsampling" and not encoder_only) or model_name == "enet-naive-upsampling-enhanced":
        for nm, c in (("de_bn_4.0", 64), ("de_bn_4.1", 64), ("de_bn_4.2", 64), ("de_bn_5.0", 16), ("de_bn_5.1", 16)):
            out[nm + "_sdrop2d_1"] = (0.1, c)
    return out


def make_dropout_masks(model_name: str, batch: int, seed: int, dtype=np.float64):
    rng = np.random.RandomState(seed)
    masks = OrderedDict()
    for name, (rate, c) in dropout_layer_rates(model_name).items():
        keep = rng.uniform(size=(batch, c)) >= rate
        masks[name] = (keep.astype(dtype) / (1.0 - rate))
    return masks


# --------------------------------------------------------------------------------------
# forward graph
# --------------------------------------------------------------------------------------

class _Ctx:
    def __init__(self, params: Dict[str, torch.Tensor], training: bool, drop_masks, collect, forced=None, storage=None,
                 bf16_weights=None):
        # storage == "bf16": every tensor the device implementation keeps in memory between kernels (convolution outputs,
        # activated outputs, block outputs, logits) and every stored gradient is rounded to bfloat16 at that point, the
        # arithmetic in between stays in this evaluation's own precision
        self.storage = storage
        # names of convolution layers whose kernel enters the convolution (forward and data gradient) rounded to bfloat16,
        # as on the bfloat16 matrix instructions; the kernel's own gradient is that of the unrounded kernel
        self.bf16_weights = set(bf16_weights or ())
        self.unrounded = {}
        self.p = params
        self.training = training
        # forced: {PReLU layer name: sign tensor (+1 / -1 / 0)} and {"<block>_other_pool2d/argmax": int64 indices}: branches
        # replayed from another evaluation instead of decided by this one's values
        self.forced = forced or {}
        self.drop = drop_masks or {}
        self.bn_stats: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.acts: Dict[str, torch.Tensor] = {} if collect else None

    def q(self, t, fwd=True, bwd=True):
        return K.round_bf16(t, fwd, bwd) if self.storage == "bf16" else t

    def keep(self, name, t):
        if self.acts is not None:
            self.acts[name] = t
        return t

    def conv(self, name, x, stride=1, dil=1, padding="same"):
        kernel = self.p[name + "/kernel"]
        if name in self.bf16_weights:
            kernel = K.round_bf16(kernel, fwd=True, bwd=False)
        y = K.conv2d(x, kernel, self.p.get(name + "/bias"), stride, dil, padding)
        if self.storage == "bf16":
            # the stored tensor is rounded, but a BatchNormalization that consumes it takes its batch statistics from the
            # convolution's unrounded accumulators (column sums in the kernel epilogue); the gradient of both uses meets in
            # y and is rounded there, as the stored gradient tensor is
            y = self.q(y, fwd=False)
            yq = y + (y.to(torch.bfloat16).to(y.dtype) - y).detach()
            self.unrounded[id(yq)] = (yq, y)
            y = yq
        return self.keep(name, y)

    def tconv(self, name, x, stride=2):
        return self.keep(name, self.q(K.conv2d_transpose(x, self.p[name + "/kernel"], self.p.get(name + "/bias"), stride)))

    def bn(self, name, x):
        g, b = self.p[name + "/gamma"], self.p[name + "/beta"]
        if self.training:
            src = self.unrounded.get(id(x), (None, x))[1]
            y, mean, var = K.batch_norm_train(x, g, b, stats_of=src)
            self.bn_stats[name] = (mean.detach(), var.detach())
        else:
            y = K.batch_norm_infer(x, g, b, self.p[name + "/moving_mean"], self.p[name + "/moving_variance"])
        return self.keep(name, y)

    def prelu(self, name, x):
        if name in self.forced:
            return self.keep(name, self.q(K.prelu_forced(x, self.p[name + "/alpha"].reshape(-1), self.forced[name])))
        return self.keep(name, self.q(K.prelu(x, self.p[name + "/alpha"].reshape(-1))))

    def sdrop(self, name, x):
        if self.training and name in self.drop:
            return K.spatial_dropout(x, self.drop[name])
        return x


def _enc_bottleneck(c: _Ctx, x, cout, p, downsample=False, asymmetric=0, dilated=0, want_idx=False):
    s = 2 if downsample else 1
    e = c.conv(p + "_proj_conv2d_1", x, stride=s, padding="valid")
    e = c.prelu(p + "_prelu_1", c.bn(p + "_bnorm_1", e))
    if asymmetric:
        e = c.conv(p + "_aconv2d_1", e)
        e = c.conv(p + "_aconv2d_2", e)
    elif dilated:
        e = c.conv(p + "_dconv2d", e, dil=dilated)
    else:
        e = c.conv(p + "_conv2d_1", e)
    e = c.prelu(p + "_prelu_2", c.bn(p + "_bnorm_2", e))
    e = c.conv(p + "_proj_conv2d_2", e, padding="valid")
    e = c.sdrop(p + "_sdrop2d_1", c.bn(p + "_bnorm_3", e))
    other, idx = x, None
    if downsample:
        fkey = p + "_other_pool2d/argmax"
        if fkey in c.forced:
            idx = c.forced[fkey]
            other = K.max_pool_forced(other, idx)
            c.keep(fkey, idx)
        elif want_idx:
            other, idx = K.max_pool_with_argmax(other)
            c.keep(p + "_other_pool2d/argmax", idx)
        else:
            other = K.max_pool_2x2(other)
        other = K.channel_zero_pad(other, cout)
    out = c.prelu(p + "_prelu_3", c.q(e + other, fwd=False))      # the gradient of the sum is a stored tensor
    return (out, idx) if want_idx and downsample else out


def _dec_bottleneck(c: _Ctx, x, cout, p, upsample, variant, idx=None):
    cin = x.shape[3]
    d = c.conv(p + "_proj_conv2d_1", x, padding="valid")
    d = c.prelu(p + "_prelu_1", c.bn(p + "_bnorm_1", d))
    if upsample and variant == "enhanced":
        d = c.conv(p + "_nn_rs_conv_conv2d", K.upsample_nn2(d))
    else:
        d = c.tconv(p + "_tconv2d_1", d) if upsample else c.conv(p + "_conv2d_1", d)
    d = c.prelu(p + "_prelu_2", c.bn(p + "_bnorm_2", d))
    d = c.conv(p + "_proj_conv2d_2", d)
    if variant in ("naive", "enhanced"):
        d = c.sdrop(p + "_sdrop2d_1", c.bn(p + "_bnorm_3", d))
    other = x
    if cin != cout or upsample:
        other = c.q(c.bn(p + "_other_bnorm_1", c.conv(p + "_other_conv2d", other)), fwd=False)
        if upsample:
            other = K.upsample_nn2(other) if variant in ("naive", "enhanced") else K.max_unpool(other, idx)
    if variant == "unpool":
        d = c.bn(p + "_bnorm_3", d)
    return c.prelu(p + "_prelu_3", c.q(d + other, fwd=False))


def classifier_head(c, x):
    """AveragePooling2D over the whole map (GlobalAveragePooling2D for the enhanced variant) -> Flatten -> Dense 'logits'
    (src/models.py:985-993, 1040-1048, 1667-1668)."""
    pooled = c.keep("avg_pool2d", x.mean(dim=(1, 2)))
    return pooled @ c.p["logits/kernel"] + c.p["logits/bias"]


def enet_forward(model_name: str, params: Dict[str, torch.Tensor], images: torch.Tensor, training: bool = True,
                 drop_masks=None, collect: bool = False, forced=None, storage=None, bf16_weights=None):
    """Returns (logits, ctx) where ctx.bn_stats holds batch (mean, biased var) per BN layer and
    ctx.acts the named intermediate tensors when collect=True."""
    model_name, encoder_only = split_name(model_name)
    if model_name in VGG_STYLE:
        c = _Ctx(params, training, drop_masks, collect, forced)
        return V.forward(model_name, c, images, encoder_only), c
    variant = {"enet-naive-upsampling": "naive", "enet-max-unpooling": "unpool",
               "enet-naive-upsampling-enhanced": "enhanced"}.get(model_name)
    if variant is None:
        raise ValueError("Unknown model name: {}".format(model_name))
    c = _Ctx(params, training, drop_masks, collect, forced, storage, bf16_weights)
    idx_stack = []
    images = c.q(images, bwd=False)
    conv = c.conv("initial_block_conv2d", images, stride=2)
    if variant in ("naive", "enhanced"):
        x = torch.cat([conv, K.max_pool_2x2(images)], dim=3)
        x = c.prelu("initial_block_prelu", c.bn("initial_block_bnorm", x))
    else:
        pooled, idx0 = K.max_pool_with_argmax(images)
        c.keep("initial_block_pool2d/argmax", idx0)
        idx_stack.append(idx0)
        x = torch.cat([conv, pooled], dim=3)
    c.keep("initial_block", x)
    want = variant == "unpool"
    r = _enc_bottleneck(c, x, 64, "en_bn_1.0", downsample=True, want_idx=want)
    x = r[0] if want else r
    if want:
        idx_stack.append(r[1])
    for i in range(4):
        x = _enc_bottleneck(c, x, 64, "en_bn_1.{}".format(i + 1))
    r = _enc_bottleneck(c, x, 128, "en_bn_2.0", downsample=True, want_idx=want)
    x = r[0] if want else r
    if want:
        idx_stack.append(r[1])
    for stage in (2, 3):
        for suffix, kw in ENC_PATTERN:
            x = _enc_bottleneck(c, x, 128, "en_bn_{}.{}".format(stage, suffix), **kw)
    c.keep("encoder", x)
    if encoder_only and variant != "enhanced":
        return classifier_head(c, x), c
    x = _dec_bottleneck(c, x, 64, "de_bn_4.0", True, variant, idx_stack.pop() if want else None)
    x = _dec_bottleneck(c, x, 64, "de_bn_4.1", False, variant)
    x = _dec_bottleneck(c, x, 64, "de_bn_4.2", False, variant)
    x = _dec_bottleneck(c, x, 16, "de_bn_5.0", True, variant, idx_stack.pop() if want else None)
    x = _dec_bottleneck(c, x, 16, "de_bn_5.1", False, variant)
    if variant == "enhanced":
        x = c.conv("final_nn_rs_conv_conv2d", K.upsample_nn2(x))
        x = c.prelu("final_nn_rs_conv_prelu", c.bn("final_nn_rs_conv_bnorm", x))
        logits = c.conv("decoder_logits" if encoder_only else "logits", x)
        if encoder_only:
            logits = classifier_head(c, logits)
    else:
        logits = c.tconv("logits", x)
    return logits, c


def block_table(model_name: str) -> "OrderedDict[str, tuple]":
    """Bottleneck name -> ("enc" | "dec", output channels, keyword arguments of its builder), network order."""
    variant = {"enet-naive-upsampling": "naive", "enet-max-unpooling": "unpool",
               "enet-naive-upsampling-enhanced": "enhanced"}[split_name(model_name)[0]]
    t: "OrderedDict[str, tuple]" = OrderedDict()
    t["en_bn_1.0"] = ("enc", 64, dict(downsample=True))
    for i in range(4):
        t["en_bn_1.{}".format(i + 1)] = ("enc", 64, {})
    t["en_bn_2.0"] = ("enc", 128, dict(downsample=True))
    for stage in (2, 3):
        for suffix, kw in ENC_PATTERN:
            t["en_bn_{}.{}".format(stage, suffix)] = ("enc", 128, dict(kw))
    for name, cout, up in (("de_bn_4.0", 64, True), ("de_bn_4.1", 64, False), ("de_bn_4.2", 64, False),
                           ("de_bn_5.0", 16, True), ("de_bn_5.1", 16, False)):
        t[name] = ("dec", cout, dict(upsample=up, variant=variant))
    return t


def block_step(model_name: str, weights, block: str, x: np.ndarray, dout: np.ndarray, drop_masks=None, forced=None,
               storage: Optional[str] = None, unpool_idx: Optional[np.ndarray] = None, dtype=torch.float64, bf16_weights=None):
    """ONE block of the network evaluated on its own: forward from the given input `x` (training phase), then the
    vector-Jacobian product with the given output gradient `dout`.  `block`: a bottleneck name, "initial_block" (x = images)
    or "head" (everything after de_bn_5.1; dout = gradient of the logits).  Used to check an implementation layer by layer
    on ITS OWN intermediate tensors, which keeps the comparison meaningful where whole-network differences compound (reduced
    precision storage).  Returns dict(out, dx, grads{key})."""
    base, _ = split_name(model_name)
    variant = {"enet-naive-upsampling": "naive", "enet-max-unpooling": "unpool", "enet-naive-upsampling-enhanced": "enhanced"}[base]
    prefixes = {"initial_block": ("initial_block",), "head": ("logits", "final_nn_rs_conv")}.get(block, (block + "_",))
    params = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in weights.items() if k.startswith(prefixes)}
    tkeys = [k for k in params if k.endswith(TRAINABLE_SUFFIXES)]
    for k in tkeys:
        params[k].requires_grad_(True)
    masks = {k: torch.tensor(v, dtype=dtype) for k, v in (drop_masks or {}).items()}
    fz = {k: torch.as_tensor(np.asarray(v)).to(torch.int64 if k.endswith("/argmax") else dtype) for k, v in (forced or {}).items()}
    c = _Ctx(params, True, masks, False, fz, storage, bf16_weights)
    xt = torch.tensor(np.asarray(x), dtype=dtype, requires_grad=block != "initial_block")
    if block == "initial_block":
        conv = c.conv("initial_block_conv2d", xt, stride=2)
        if variant in ("naive", "enhanced"):
            out = c.prelu("initial_block_prelu", c.bn("initial_block_bnorm", torch.cat([conv, K.max_pool_2x2(xt)], dim=3)))
        else:
            out = torch.cat([conv, K.max_pool_with_argmax(xt)[0]], dim=3)
    elif block == "head":
        if variant == "enhanced":
            h = c.conv("final_nn_rs_conv_conv2d", K.upsample_nn2(xt))
            h = c.prelu("final_nn_rs_conv_prelu", c.bn("final_nn_rs_conv_bnorm", h))
            out = c.conv("logits", h)
        else:
            out = c.tconv("logits", xt)
    else:
        kind, cout, kw = block_table(model_name)[block]
        if kind == "enc":
            out = _enc_bottleneck(c, xt, cout, block, **kw)
        else:
            idx = None if unpool_idx is None else torch.as_tensor(np.asarray(unpool_idx)).to(torch.int64)
            out = _dec_bottleneck(c, xt, cout, block, kw["upsample"], kw["variant"], idx)
    wanted = [params[k] for k in tkeys] + ([xt] if xt.requires_grad else [])
    grads = torch.autograd.grad(out, wanted, grad_outputs=torch.tensor(np.asarray(dout), dtype=dtype), allow_unused=True)
    gd = {k: (g.detach().numpy() if g is not None else np.zeros(tuple(params[k].shape))) for k, g in zip(tkeys, grads)}
    dx = grads[-1].detach().numpy() if xt.requires_grad else None
    if dx is not None and storage == "bf16":
        dx = torch.from_numpy(dx).to(torch.bfloat16).to(dtype).numpy()      # the block's input gradient is a stored tensor
    return dict(out=out.detach().numpy(), dx=dx, grads=gd)


TRAINABLE_SUFFIXES = ("/kernel", "/bias", "/gamma", "/beta", "/alpha")


def trainable_keys(model_name: str, num_classes: int) -> List[str]:
    return [k for k, _, _, _ in weight_spec(model_name, num_classes) if k.endswith(TRAINABLE_SUFFIXES)]


def train_step(model_name: str, weights: "OrderedDict[str, np.ndarray]", images: np.ndarray, labels: np.ndarray,
               pixel_weights: np.ndarray, drop_masks=None, dtype=torch.float64, collect=False,
               adam: Optional[dict] = None, adam_state: Optional[dict] = None, ignore_classes=(0,),
               mean_teacher: Optional[dict] = None, superpixel: Optional[dict] = None, classification: bool = False,
               forced: Optional[dict] = None, storage: Optional[str] = None, bf16_weights=None):
    """One supervised training step exactly as the reference assembles it
    (src/trainers.py:2436-2440: total = 1*loss + 0*mean(0*logits)):
    forward (training phase) -> weighted CE -> gradients -> BN moving-average updates -> optional Adam.
    With mean_teacher = dict(num_unlabeled, mt_logits, coefficient) the loss is the mean-teacher lambda loss
    (src/losses.py:377-436); with superpixel = dict(num_unlabeled, coefficient) the superpixel lambda loss (:439-492); with
    both the combined one (:495-572).  The metrics see zero labels on the unlabeled rows (src/generators.py:1764-1766).
    Returns dict(loss, logits, grads{key}, new_weights{key}, cm, acts)."""
    num_classes = weights["logits/bias"].shape[0]
    params = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    tkeys = trainable_keys(model_name, num_classes)
    for k in tkeys:
        params[k].requires_grad_(True)
    x = torch.tensor(images, dtype=dtype)
    masks = {k: torch.tensor(v, dtype=dtype) for k, v in (drop_masks or {}).items()}
    fz = None
    if forced:
        fz = {k: torch.as_tensor(np.asarray(v)).to(torch.int64 if k.endswith("/argmax") else dtype) for k, v in forced.items()}
    logits, ctx = enet_forward(model_name, params, x, True, masks, collect, fz, storage, bf16_weights)
    if classification:      # labels / pixel_weights are [B, classes] float arrays (src/losses.py:636-652)
        lab = torch.tensor(labels, dtype=dtype)
        w = torch.tensor(pixel_weights, dtype=dtype)
        loss = K.classification_weighted_categorical_crossentropy(lab, logits, w)
    else:
        lab = torch.tensor(labels.astype(np.int64))
        w = torch.tensor(pixel_weights, dtype=dtype)
    if classification:
        pass
    elif mean_teacher is None and superpixel is None:
        loss = K.segmentation_cce_lambda_loss(logits, lab, w, 0)
    else:
        nu = int((mean_teacher or superpixel)["num_unlabeled"])
        if mean_teacher is not None:
            mt = torch.tensor(mean_teacher["mt_logits"], dtype=dtype)
        if mean_teacher is not None and superpixel is not None:
            loss = K.segmentation_mean_teacher_superpixel_lambda_loss(logits, lab, w, nu, mt, float(mean_teacher["coefficient"]),
                                                                      float(superpixel["coefficient"]))
        elif mean_teacher is not None:
            loss = K.segmentation_mean_teacher_lambda_loss(logits, lab, w, nu, mt, float(mean_teacher["coefficient"]))
        else:
            loss = K.segmentation_superpixel_lambda_loss(logits, lab, w, nu, float(superpixel["coefficient"]))
        if nu:
            labels = labels.copy()
            labels[labels.shape[0] - nu:] = 0
    if collect:
        logits.retain_grad()
    grads = torch.autograd.grad(loss, [params[k] for k in tkeys] + ([logits] if collect else []), allow_unused=True)
    gdict = {k: (g.detach().numpy() if g is not None else np.zeros_like(weights[k])) for k, g in zip(tkeys, grads)}
    new_w = OrderedDict((k, np.array(v)) for k, v in weights.items())
    for name, (mean, var) in ctx.bn_stats.items():
        new_w[name + "/moving_mean"] = K.bn_moving_update(torch.tensor(weights[name + "/moving_mean"], dtype=dtype), mean).numpy()
        new_w[name + "/moving_variance"] = K.bn_moving_update(torch.tensor(weights[name + "/moving_variance"], dtype=dtype), var).numpy()
    if adam is not None:
        st = adam_state if adam_state is not None else {"iterations": 0, "m": {}, "v": {}}
        ps = [torch.tensor(weights[k], dtype=dtype) for k in tkeys]
        gs = [torch.tensor(gdict[k], dtype=dtype) for k in tkeys]
        ms = [st["m"].get(k, torch.zeros_like(p)) for k, p in zip(tkeys, ps)]
        vs = [st["v"].get(k, torch.zeros_like(p)) for k, p in zip(tkeys, ps)]
        K.adam_step(ps, gs, ms, vs, st["iterations"], **adam)
        for k, p, m, v in zip(tkeys, ps, ms, vs):
            new_w[k] = p.numpy()
            st["m"][k], st["v"][k] = m, v
        st["iterations"] += 1
    cm = None if classification else K.confusion_matrix_np(labels, logits.detach().numpy(), num_classes, ignore_classes)
    out = dict(loss=float(loss.detach()), logits=logits.detach().numpy(), grads=gdict, new_weights=new_w, cm=cm)
    if collect:
        out["acts"] = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in ctx.acts.items()}
        out["dlogits"] = grads[-1].detach().numpy()
    return out


def synthetic_batch(batch: int, height: int, width: int, num_classes: int = 24, seed: int = 14874, block: int = 32,
                    class_weights: Optional[np.ndarray] = None):
    """Synthetic inputs of the shape the reference's generator emits (src/generators.py:1742-1766):
    images f32 ~ N(0,1); labels int32 constant over `block`x`block` tiles; per-pixel weights = w[labels]."""
    rng = np.random.RandomState(seed)
    images = rng.standard_normal((batch, height, width, 3)).astype(np.float32)
    bh, bw = -(-height // block), -(-width // block)
    coarse = rng.randint(0, num_classes, size=(batch, bh, bw)).astype(np.int32)
    labels = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :height, :width].copy()
    if class_weights is None:
        class_weights = np.ones(num_classes, dtype=np.float32)
        class_weights[0] = 0.0
    weights = class_weights.astype(np.float32)[labels]
    return images, labels, weights
