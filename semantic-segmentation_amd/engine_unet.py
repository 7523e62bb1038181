"""Layer graph of the U-Net and SegNet-Basic builders (src/models.py:771-1012) on the execution engine.

Both networks are chains of the reference's `get_convolution_block` (src/models.py:828-878):
    ZeroPadding2D(1) -> Conv2D(3x3, 'valid') -> BatchNormalization(momentum 0.1) -> [LeakyReLU(0.1)] -> SpatialDropout2D(0.1)
which runs here as TWO passes over the activation: the implicit-GEMM convolution (zero padding is the kernel's bounds
check, never a padded copy) and one fused BN + LeakyReLU + dropout pass; the backward mirrors it.
UpSampling2D + concatenate (src/models.py:920-927) never materialise separately: the upsampling kernel and the skip copy
write the two channel slices of one buffer, and the gradient of the skip slice is added straight from the decoder's
gradient buffer when the encoder block runs its backward.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ._lib import lib
from .engine import (ACT_LEAKY, ACT_NONE, BatchNorm, Block, ClassifierHead, Conv, GraphBase, ParamStore,
                     RunCfg, _ptr)

DROP_RATE = 0.1     # src/models.py:841
LEAKY_ALPHA = 0.1   # src/models.py:840


class ConvUnit:
    """One `get_convolution_block`.  Keeps the raw convolution output `y` (BN/activation are recomputed from it in the
    backward pass) and the block output `out`."""

    def __init__(self, store: ParamStore, name, cin, cout, use_bias, use_activation):
        self.name, self.cin, self.cout = name, cin, cout
        self.store = store
        self.conv = Conv(store, name, 3, 3, cin, cout, padding="same", bias=use_bias)   # pad 1 + valid == SAME for 3x3/s1
        self.bn = BatchNorm(store, name + "_normalization", cout)
        self.act = ACT_LEAKY if use_activation else ACT_NONE
        self.drop_name = name + "_dropout"
        self.drop_mask = None
        self.needs_dx = True

    def plan(self, n, h, w, dev, x_ld=None):
        f32 = dict(dtype=self.store.act_dtype, device=dev)      # activation type
        self.n, self.h, self.w, self.npix = n, h, w, n * h * w
        self.conv.plan(n, h, w, x_ld=x_ld)
        self.bn.plan(dev)
        self.y = torch.empty(n, h, w, self.cout, **f32)
        self.out = torch.empty_like(self.y)
        self.dy = torch.empty_like(self.y)
        self.dx = torch.empty(n, h, w, self.cin, **f32) if self.needs_dx else None

    def fwd(self, x, cfg: RunCfg):
        L = self.store.L
        bn = self.bn
        self.x = x
        if not cfg.training and not cfg.use_dropout:      # inference phase: BN + activation in the convolution's epilogue
            self.conv.fwd_bn_act_infer(x, self.out, bn, cfg.st, act=self.act, slope=LEAKY_ALPHA)
            return self.out
        self.conv.fwd_bn(x, self.y, bn, self.npix, cfg.training, cfg.update_moving, cfg.unbiased_moving, cfg.ws, cfg.st)
        drop = self.drop_mask if cfg.use_dropout else None
        L.bn_act_drop_fwd(_ptr(self.y), self.cout, _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma), _ptr(bn.beta), self.act,
                          None, LEAKY_ALPHA, _ptr(drop), self.h * self.w, _ptr(self.out), self.cout, self.npix, self.cout,
                          cfg.st)
        return self.out

    def bn_bwd(self, dout_ptr, dout_ld, cfg: RunCfg):
        """dout: gradient w.r.t. `out` (raw pointer + pixel stride, it may be a channel slice) -> self.dy (gradient of
        the raw convolution output) and the BN parameter gradients."""
        L = self.store.L
        bn = self.bn
        drop = self.drop_mask if cfg.use_dropout else None
        L.bn_act_drop_bwd(dout_ptr, dout_ld, _ptr(self.y), self.cout, _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma),
                          _ptr(bn.beta), self.act, None, LEAKY_ALPHA, _ptr(drop), self.h * self.w, _ptr(self.dy),
                          _ptr(bn.dgamma), _ptr(bn.dbeta), None, self.npix, self.cout, _ptr(cfg.ws), cfg.ws.numel() * 4,
                          cfg.st)

    def conv_bwd(self, cfg: RunCfg, producer: "Optional[ConvUnit]" = None):
        """Weight gradient and data gradient of the convolution from self.dy.  With `producer` (the unit whose output is
        this unit's input) the data gradient and the producer's BN backward run as one fused call and land in
        producer.dy; returns dx otherwise (None when the input needs no gradient)."""
        cfg.wgrad(self.conv, self.x, self.dy)
        if producer is not None:
            p = producer
            drop = p.drop_mask if cfg.use_dropout else None
            self.conv.bwd_data_bn(self.dy, self.dx, p.bn, p.y, p.cout, p.dy, p.npix, cfg.ws, cfg.st, act=p.act,
                                  slope=LEAKY_ALPHA, drop=drop, hw=p.h * p.w)
            return None
        if self.needs_dx:
            self.conv.bwd_data(self.dy, self.dx, cfg.st, ws=cfg.ws)
        return self.dx


class UnitBlock(Block):
    """Shared plumbing of the blocks below: a run of ConvUnits."""

    units: List[ConvUnit]

    def convs(self):
        return [u.conv for u in self.units]

    def dropouts(self):
        return [(u.drop_name, DROP_RATE, u.cout) for u in self.units]

    def bind_dropout(self, views: Dict[str, torch.Tensor]):
        for u in self.units:
            u.drop_mask = views[u.drop_name]

    def _fwd_units(self, x, cfg):
        for u in self.units:
            x = u.fwd(x, cfg)
        return x

    def _bwd_units(self, d: torch.Tensor, cfg):
        us = self.units
        us[-1].bn_bwd(_ptr(d), us[-1].cout, cfg)
        for k in range(len(us) - 1, 0, -1):
            us[k].conv_bwd(cfg, producer=us[k - 1])      # also produces us[k-1].dy
        return us[0].conv_bwd(cfg)


class EncoderBlock(UnitBlock):
    """`get_encoder_block` (src/models.py:881-907): n convolution blocks, then MaxPooling2D(2x2).  `skip` (the pre-pool
    tensor) is what the paired decoder block concatenates."""

    def __init__(self, store, prefix, n_conv, cin, cout, first=False):
        self.name = prefix
        self.store = store
        self.units = []
        c = cin
        for i in range(n_conv):
            self.units.append(ConvUnit(store, "{}_conv{}".format(prefix, i + 1), c, cout, True, True))
            c = cout
        if first:
            self.units[0].needs_dx = False     # the images need no gradient
        self.cout = cout
        self.consumer: Optional["DecoderBlock"] = None

    def plan(self, n, h, w, dev):
        for u in self.units:
            u.plan(n, h, w, dev)
        self.n, self.h, self.w = n, h, w
        c = self.cout
        self.pooled = torch.empty(n, h // 2, w // 2, c, dtype=self.store.act_dtype, device=dev)
        self.idx = torch.empty(n, h // 2, w // 2, c, dtype=torch.int32, device=dev)
        self.dskip = torch.empty(n, h, w, c, dtype=self.store.act_dtype, device=dev)
        return (h // 2, w // 2, c)

    @property
    def skip(self):
        return self.units[-1].out

    def fwd(self, x, cfg):
        s = self._fwd_units(x, cfg)
        self.store.L.maxpool_argmax_fwd(_ptr(s), self.n, self.h, self.w, self.cout, _ptr(self.pooled), self.cout, _ptr(self.idx),
                                 None, cfg.st)
        return self.pooled

    def bwd(self, dpooled, cfg):
        L = self.store.L
        c = self.cout
        L.maxpool_argmax_bwd(_ptr(dpooled), c, _ptr(self.idx), self.n, self.h, self.w, c, _ptr(self.dskip), 0, cfg.st)
        if self.consumer is not None:
            # gradient of the concatenated copy: channel slice [c_up, c_up + c) of the decoder's input gradient
            dc = self.consumer
            L.copy_ld(_ptr(dc.dcat) + self.store.esize * dc.c_up, dc.c_cat, _ptr(self.dskip), c, self.n * self.h * self.w, c, 1, cfg.st)
        return self._bwd_units(self.dskip, cfg)


class ConnectingBlock(UnitBlock):
    """The two 1024-wide convolution blocks between the U-Net paths (src/models.py:801-809)."""

    def __init__(self, store, names_widths, cin):
        self.name = "connecting_block"
        self.store = store
        self.units = []
        c = cin
        for name, f in names_widths:
            self.units.append(ConvUnit(store, name, c, f, True, True))
            c = f
        self.cout = c

    def plan(self, n, h, w, dev):
        for u in self.units:
            u.plan(n, h, w, dev)
        return (h, w, self.cout)

    def fwd(self, x, cfg):
        return self._fwd_units(x, cfg)

    def bwd(self, d, cfg):
        return self._bwd_units(d, cfg)


class DecoderBlock(UnitBlock):
    """`get_decoder_block` (src/models.py:910-943): UpSampling2D(2) -> concatenate([up, skip]) -> n convolution blocks
    without bias and without activation."""

    def __init__(self, store, prefix, n_conv, c_up, skip_block: EncoderBlock, cout):
        self.name = prefix
        self.store = store
        self.c_up, self.c_skip = c_up, skip_block.cout
        self.c_cat = c_up + skip_block.cout
        self.skip_block = skip_block
        skip_block.consumer = self
        self.units = []
        c = self.c_cat
        for i in range(n_conv):
            self.units.append(ConvUnit(store, "{}_conv{}".format(prefix, i + 1), c, cout, False, False))
            c = cout
        self.cout = cout

    def plan(self, n, h, w, dev):
        self.n, self.h, self.w = n, h, w
        H, W = 2 * h, 2 * w
        assert (self.skip_block.h, self.skip_block.w) == (H, W)
        self.cat = torch.empty(n, H, W, self.c_cat, dtype=self.store.act_dtype, device=dev)
        for u in self.units:
            u.plan(n, H, W, dev)
        self.dcat = self.units[0].dx
        self.dx = torch.empty(n, h, w, self.c_up, dtype=self.store.act_dtype, device=dev)
        return (H, W, self.cout)

    def fwd(self, x, cfg):
        L = self.store.L
        L.upsample_nn2x_fwd(_ptr(x), self.n, self.h, self.w, self.c_up, _ptr(self.cat), self.c_cat, cfg.st)
        L.copy_ld(_ptr(self.skip_block.skip), self.c_skip, _ptr(self.cat) + self.store.esize * self.c_up, self.c_cat,
                  self.n * 4 * self.h * self.w, self.c_skip, 0, cfg.st)
        return self._fwd_units(self.cat, cfg)

    def bwd(self, d, cfg):
        self._bwd_units(d, cfg)
        self.store.L.upsample_nn2x_bwd(_ptr(self.dcat), self.c_cat, self.n, self.h, self.w, self.c_up, _ptr(self.dx), cfg.st)
        return self.dx


class PointwiseHead(Block):
    """Conv2D(num_classes, 1x1) named 'logits' (src/models.py:823, 1010)."""

    def __init__(self, store, cin, num_classes):
        self.name = "logits"
        self.store = store
        self.conv = Conv(store, "logits", 1, 1, cin, num_classes, padding="valid", bias=True)
        self.cin, self.num_classes = cin, num_classes

    def convs(self):
        return [self.conv]

    def plan(self, n, h, w, dev):
        self.conv.plan(n, h, w)
        self.out = torch.empty(n, h, w, self.num_classes, dtype=self.store.act_dtype, device=dev)
        self.dout = torch.empty_like(self.out)
        self.dx = torch.empty(n, h, w, self.cin, dtype=self.store.act_dtype, device=dev)
        return (h, w, self.num_classes)

    def fwd(self, x, cfg):
        self.x = x
        self.conv.fwd(x, self.out, cfg.st)
        return self.out

    def bwd(self, dout, cfg):
        cfg.wgrad(self.conv, self.x, dout)
        self.conv.bwd_data(dout, self.dx, cfg.st, ws=cfg.ws)
        return self.dx


UNET = dict(
    enc=[("encoder_block1", 2, 64), ("encoder_block2", 2, 128), ("encoder_block3", 2, 256), ("encoder_block4", 2, 512)],
    connect=[("connecting_block_conv1", 1024), ("connecting_block_conv2", 1024)],
    dec=[("decoder_block1", 2, 512, 3), ("decoder_block2", 2, 256, 2), ("decoder_block3", 2, 128, 1),
         ("decoder_block4", 2, 64, 0)])                                      # src/models.py:793-817
SEGNET = dict(
    enc=[("encoder_block1", 2, 64), ("encoder_block2", 2, 128), ("encoder_block3", 3, 256), ("encoder_block4", 3, 512)],
    connect=[],
    dec=[("decoder_block2", 3, 512, 3), ("decoder_block3", 3, 256, 2), ("decoder_block4", 2, 128, 1),
         ("decoder_block5", 2, 64, 0)])                                      # src/models.py:979-1004


class UNetSegNetGraph(GraphBase):
    """`unet` / `segnet`: four encoder blocks, (U-Net) the connecting block, four decoder blocks, 1x1 logits."""

    def __init__(self, kind: str, num_classes: int, device="cuda", encoder_only: bool = False, act_dtype=torch.float32):
        assert kind in ("unet", "segnet")
        assert not (encoder_only and kind == "unet")      # the reference has no unet-encoder-only
        topo = UNET if kind == "unet" else SEGNET
        self.variant, self.num_classes, self.encoder_only = kind, num_classes, encoder_only
        self.device = torch.device(device)
        st = self.store = ParamStore(act_dtype)
        blocks: List[Block] = []
        encs: List[EncoderBlock] = []
        c = 3
        for i, (prefix, n, f) in enumerate(topo["enc"]):
            encs.append(EncoderBlock(st, prefix, n, c, f, first=(i == 0)))
            c = f
        blocks.extend(encs)
        if encoder_only:      # src/models.py:985-993: average pooling over pool4 -> Flatten -> Dense
            blocks.append(ClassifierHead(st, c, num_classes))
            self._finish_init(blocks)
            return
        if topo["connect"]:
            cb = ConnectingBlock(st, topo["connect"], c)
            blocks.append(cb)
            c = cb.cout
        for prefix, n, f, skip in topo["dec"]:
            blocks.append(DecoderBlock(st, prefix, n, c, encs[skip], f))
            c = f
        blocks.append(PointwiseHead(st, c, num_classes))
        self._finish_init(blocks)

    def _plan_blocks(self, n, h, w, dev):
        hh, ww = h, w
        for b in self.blocks:
            hh, ww, _ = b.plan(n, hh, ww, dev)
