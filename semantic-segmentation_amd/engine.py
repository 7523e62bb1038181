"""Execution engine: the layer graph of an ENet-style encoder-decoder laid out for one MI355X.

Design (one process per GPU):
  * all trainable tensors live in ONE flat fp32 arena `theta` (16-byte aligned segments, Keras creation order);
    gradients, Adam/SGD slots mirror it, so the optimizer is one kernel and the data-parallel all-reduce is a
    bucketed reduce over a contiguous buffer;
  * BatchNormalization moving statistics live in a second arena `state`;
  * activations are NHWC fp32 device buffers allocated once per input shape; every arithmetic op is a call into
    libssmi.so (HIP, gfx950) on the current HIP stream, so a whole step can be captured into a HIP graph;
  * blocks are the fused units of the reference's builders (initial block, encoder / decoder bottleneck, logits head):
    concat / zero-pad / upsample / unpool / add / PReLU / dropout are never materialised on their own.

Reference topology: src/models.py:1054-1319 (ENetNaiveUpsampling), 1360-1628 (ENetMaxUnpooling).
"""
from __future__ import annotations

import os
import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ._lib import BF16, F32, BnParams, ConvDesc, ExpandDesc, ExpandNextApply, MergeDesc, Premerge, lib, lib_for

BN_EPS = 1e-3       # Keras BatchNormalization default
BN_MOMENTUM = 0.1   # src/models.py:1064 (weight of the OLD moving value)
ACT_NONE, ACT_PRELU, ACT_LEAKY = 0, 1, 2


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def same_pad(size: int, k: int, stride: int, dil: int = 1) -> Tuple[int, int]:
    """TF SAME rule -> (out, pad_before)."""
    k_eff = (k - 1) * dil + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + k_eff - size, 0)
    return out, total // 2


# ----------------------------------------------------------------------------------------------------------
# parameter bookkeeping
# ----------------------------------------------------------------------------------------------------------

class ParamSpec:
    __slots__ = ("key", "shape", "trainable", "init", "fan_in", "offset", "size")

    def __init__(self, key, shape, trainable, init, fan_in=0):
        self.key, self.shape, self.trainable, self.init, self.fan_in = key, tuple(shape), trainable, init, fan_in
        self.size = int(np.prod(shape))
        self.offset = -1


class ParamStore:
    """Flat arenas + named views.  Order of registration = Keras layer-creation order (weight order of
    model.get_weights(): per layer kernel, bias / gamma, beta, moving_mean, moving_variance / alpha)."""

    def __init__(self, act_dtype=torch.float32):
        # element type of the ACTIVATION tensors of the network built on this store (float32 or bfloat16); parameters,
        # gradients of parameters, statistics and optimizer state are float32 either way
        if act_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("activation dtype must be torch.float32 or torch.bfloat16, got {}".format(act_dtype))
        self.act_dtype = act_dtype
        self.dtype_code = BF16 if act_dtype == torch.bfloat16 else F32
        self.esize = 2 if act_dtype == torch.bfloat16 else 4      # bytes per activation element
        self._L = None
        self.specs: "OrderedDict[str, ParamSpec]" = OrderedDict()
        self.n_train = 0
        self.n_state = 0
        self.theta = self.grad = self.state = None
        self.views: Dict[str, torch.Tensor] = {}
        self.gviews: Dict[str, torch.Tensor] = {}

    @property
    def L(self):
        """The C-ABI library as this network's activation type sees it (bfloat16 twins where they exist)."""
        if self._L is None:
            self._L = lib_for(self.dtype_code)
        return self._L

    def add(self, key, shape, trainable=True, init="zeros", fan_in=0):
        if key in self.specs:
            raise ValueError("duplicate weight name: {}".format(key))
        s = ParamSpec(key, shape, trainable, init, fan_in)
        if trainable:
            s.offset = self.n_train
            self.n_train += (s.size + 3) // 4 * 4
        else:
            s.offset = self.n_state
            self.n_state += (s.size + 3) // 4 * 4
        self.specs[key] = s
        return s

    def materialise(self, device):
        self.theta = torch.zeros(max(self.n_train, 4), dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.theta)
        self.state = torch.zeros(max(self.n_state, 4), dtype=torch.float32, device=device)
        # per-layer (mean, 1/std) used by the BN kernels, laid out like `state`: in the inference phase one launch fills all
        # of it from the moving statistics, in the training phase every layer's statistics kernel writes its own slice
        self.bn_stats = torch.zeros_like(self.state)
        mask = np.zeros(self.state.numel(), dtype=np.uint8)
        for k, s in self.specs.items():
            if not s.trainable and k.endswith("/moving_variance"):
                mask[s.offset:s.offset + s.size] = 1
        self.var_mask = torch.from_numpy(mask).to(device)
        for k, s in self.specs.items():
            arena = self.theta if s.trainable else self.state
            self.views[k] = arena[s.offset:s.offset + s.size].view(s.shape)
            if s.trainable:
                self.gviews[k] = self.grad[s.offset:s.offset + s.size].view(s.shape)

    def trainable_keys(self) -> List[str]:
        return [k for k, s in self.specs.items() if s.trainable]

    def init_values(self, seed: int):
        """Keras default initialisers: he_normal for bottleneck/decoder kernels, glorot_uniform for the initial conv,
        zeros for biases/beta/alpha/moving_mean, ones for gamma/moving_variance."""
        rng = np.random.RandomState(seed)
        for k, s in self.specs.items():
            if s.init == "he_normal":
                v = rng.standard_normal(s.shape) * np.sqrt(2.0 / s.fan_in)
            elif s.init == "glorot_uniform" and len(s.shape) == 2:      # Dense
                lim = np.sqrt(6.0 / (s.shape[0] + s.shape[1]))
                v = rng.uniform(-lim, lim, size=s.shape)
            elif s.init == "glorot_uniform":
                kh, kw, cin, cout = s.shape
                lim = np.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
                v = rng.uniform(-lim, lim, size=s.shape)
            elif s.init == "ones":
                v = np.ones(s.shape)
            else:
                v = np.zeros(s.shape)
            self.views[k].copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))


# ----------------------------------------------------------------------------------------------------------
# primitive layers (thin objects around one or two C-ABI calls)
# ----------------------------------------------------------------------------------------------------------

class Conv:
    """Keras Conv2D.  `transpose=True`: Keras Conv2DTranspose(padding='same', strides=stride); then cin/cout are the
    transposed layer's own in/out channel counts and the kernel has the Keras layout (kh, kw, out, in)."""

    def __init__(self, store: ParamStore, name, kh, kw, cin, cout, stride=1, dil=1, padding="same", bias=True,
                 transpose=False, init="he_normal", kernel_shape=None):
        self.name, self.kh, self.kw, self.cin, self.cout = name, kh, kw, cin, cout
        self.stride, self.dil, self.padding, self.transpose = stride, dil, padding, transpose
        if kernel_shape is not None:      # a Keras Dense kernel (cin, cout): the same memory as a 1x1 convolution kernel
            assert int(np.prod(kernel_shape)) == kh * kw * cin * cout and not transpose
            store.add(name + "/kernel", kernel_shape, True, init, kh * kw * cin)
        elif transpose:
            store.add(name + "/kernel", (kh, kw, cout, cin), True, init, kh * kw * cin)
        else:
            store.add(name + "/kernel", (kh, kw, cin, cout), True, init, kh * kw * cin)
        self.has_bias = bias
        if bias:
            store.add(name + "/bias", (cout,), True, "zeros")
        self.store = store
        self.desc = None
        self.ws_bytes = 0
        self._ws_view = None      # this layer's own weight-gradient workspace when the strip reductions are grouped (GraphBase.plan)
        self.pack_usage = 0       # bf16 storage: bit 0 / 1 = forward / data gradient can run from a bfloat16 kernel pack
        self.uses_rot = False     # the data gradient can use a pre-rotated kernel matrix (decided by the library in plan())
        self.prebn_ok = False
        self.premerge_ok = False
        self.wrot = None          # that matrix: a view of the network's rotated arena, refreshed once per step

    def plan(self, n, h, w, x_ld=None, y_ld=None):
        """(h, w) is the spatial size of this layer's INPUT.  Returns the output (h, w)."""
        L = self.store.L
        d = ConvDesc()
        d.dtype = self.store.dtype_code
        if not self.transpose:
            if self.padding == "same":
                ho, pt = same_pad(h, self.kh, self.stride, self.dil)
                wo, pl = same_pad(w, self.kw, self.stride, self.dil)
            else:
                ho = (h - ((self.kh - 1) * self.dil + 1)) // self.stride + 1
                wo = (w - ((self.kw - 1) * self.dil + 1)) // self.stride + 1
                pt = pl = 0
            d.n, d.h, d.w, d.cin, d.ho, d.wo, d.cout = n, h, w, self.cin, ho, wo, self.cout
            d.x_ld = x_ld or self.cin
            d.y_ld = y_ld or self.cout
            out_hw = (ho, wo)
            d.kh, d.kw, d.stride, d.dil, d.pad_t, d.pad_l = self.kh, self.kw, self.stride, self.dil, pt, pl
            self.ws_bytes = max(L.conv2d_bwd_weight_workspace_bytes(C.byref(d)),
                                L.conv2d_fwd_bn_stats_workspace_bytes(C.byref(d)),
                                L.conv2d_bwd_data_bn_bwd_workspace_bytes(C.byref(d)),
                                L.conv2d_bwd_data_workspace_bytes(C.byref(d)))
            self.bwd_data_ws_bytes = L.conv2d_bwd_data_workspace_bytes(C.byref(d))
            self.uses_rot = bool(L.conv2d_bwd_data_uses_rotated(C.byref(d)))
            # forward pass and weight gradient can apply the BatchNormalization + PReLU in front of this layer to their
            # operand themselves (ssmi_conv2d_*_prebn)
            self.prebn_ok = bool(lib().conv2d_prebn_supported(C.byref(d), 1 if self.has_bias else 0)) and \
                not os.environ.get("SSMI_NO_PREBN")
            # the data gradient can form the output of a bottleneck's merge backward on load (ssmi_conv2d_bwd_data_bn_bwd_premerge)
            # (opt-in through SSMI_PREMERGE=1: measured slower - the data gradient then reads two tensors instead of one while
            # the stored tensor is still written for the weight gradient)
            self.premerge_ok = bool(os.environ.get("SSMI_PREMERGE")) and bool(lib().conv2d_bwd_data_premerge_supported(C.byref(d)))
            self.pack_usage = int(L.conv2d_pack_usage(C.byref(d))) if not os.environ.get("SSMI_NO_BF16_MFMA") else 0
        else:
            # the forward convolution this layer is the gradient of: [n, h*s, w*s, cout_t] -> [n, h, w, cin_t]
            H, W = h * self.stride, w * self.stride
            ho, pt = same_pad(H, self.kh, self.stride, 1)
            wo, pl = same_pad(W, self.kw, self.stride, 1)
            assert (ho, wo) == (h, w)
            d.n, d.h, d.w, d.cin, d.ho, d.wo, d.cout = n, H, W, self.cout, h, w, self.cin
            d.x_ld = y_ld or self.cout   # tconv output is the underlying conv's x
            d.y_ld = x_ld or self.cin
            out_hw = (H, W)
            d.kh, d.kw, d.stride, d.dil, d.pad_t, d.pad_l = self.kh, self.kw, self.stride, 1, pt, pl
            self.ws_bytes = L.conv2d_transpose_bwd_weight_workspace_bytes(C.byref(d))
            self.bwd_data_ws_bytes = 0
        self.desc = d
        self._dref = C.byref(d)
        v, g = self.store.views, self.store.gviews
        self.w, self.dw = v[self.name + "/kernel"], g[self.name + "/kernel"]
        self.b = v[self.name + "/bias"] if self.has_bias else None
        self.db = g[self.name + "/bias"] if self.has_bias else None
        return out_hw

    def fwd(self, x, y, st):
        L = self.store.L
        if self.transpose:
            L.conv2d_transpose_fwd(self._dref, _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(y), st)
        else:
            L.conv2d_fwd(self._dref, _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(y), st)

    def fwd_bn(self, x, y, bn: "BatchNorm", npix, training, update_moving, unbiased, ws, st):
        """Convolution followed by the statistics of the BatchNormalization that consumes it.  In the training phase
        the column sums come out of the convolution kernel's epilogue (no second pass over y)."""
        if self.transpose or not training:
            self.fwd(x, y, st)
            bn.stats(y, npix, self.cout, training, update_moving, unbiased, ws, st)
            return
        self.store.L.conv2d_fwd_bn_stats(self._dref, _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(y), BN_EPS, BN_MOMENTUM,
                                  1 if unbiased else 0, _ptr(bn.mm) if update_moving else None,
                                  _ptr(bn.mv) if update_moving else None, _ptr(bn.mean), _ptr(bn.invstd), _ptr(ws),
                                  ws.numel() * 4, st)

    def fwd_bn_prebn(self, x_raw, pre: "BatchNorm", y, bn: "BatchNorm", update_moving, unbiased, ws, st):
        """fwd_bn on act(BN_pre(x_raw)) without that tensor being stored (training phase; `prebn_ok` layers)."""
        lib().conv2d_fwd_bn_stats_prebn(self._dref, _ptr(x_raw), pre._pref, _ptr(pre.alpha), _ptr(self.w), _ptr(self.b), _ptr(y),
                                        BN_EPS, BN_MOMENTUM, 1 if unbiased else 0, _ptr(bn.mm) if update_moving else None,
                                        _ptr(bn.mv) if update_moving else None, _ptr(bn.mean), _ptr(bn.invstd), _ptr(ws),
                                        ws.numel() * 4, st)

    def bwd_weight_prebn(self, x_raw, pre: "BatchNorm", dy, ws, st):
        lib().conv2d_bwd_weight_prebn(self._dref, _ptr(x_raw), pre._pref, _ptr(pre.alpha), _ptr(dy), _ptr(self.dw), _ptr(self.db),
                                      _ptr(ws), ws.numel() * 4, st)

    def fwd_bn_act_infer(self, x, y, bn: "BatchNorm", st, act=None, slope=0.0):
        """Inference phase: convolution + BatchNormalization (statistics already in bn.mean / bn.invstd) + activation in
        one launch; `y` receives the ACTIVATED output."""
        assert not self.transpose
        if act is None:
            act = ACT_PRELU if bn.alpha is not None else ACT_NONE
        self.store.L.conv2d_fwd_bn_act_infer(self._dref, _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(bn.mean), _ptr(bn.invstd),
                                      _ptr(bn.gamma), _ptr(bn.beta), act, _ptr(bn.alpha), slope, _ptr(y), st)

    def fwd_bn_add_act_infer(self, x, y, bn: "BatchNorm", skip, skip_ld, alpha, st):
        """Inference phase: y = PReLU_alpha(BN(conv(x)) + skip) in one launch (the tail of a regular bottleneck)."""
        assert not self.transpose
        self.store.L.conv2d_fwd_bn_add_act_infer(self._dref, _ptr(x), _ptr(self.w), _ptr(self.b), _ptr(bn.mean), _ptr(bn.invstd),
                                          _ptr(bn.gamma), _ptr(bn.beta), _ptr(skip), skip_ld, ACT_PRELU, _ptr(alpha), 0.0,
                                          _ptr(y), st)

    def bwd_data(self, dy, dx, st, add=None, ws=None):
        """ws: the main stream's workspace; wide layers build their rotated kernel matrix in it (see ssmi_conv2d_bwd_data_ws)."""
        L = self.store.L
        if self.transpose:
            assert add is None
            L.conv2d_transpose_bwd_data(self._dref, _ptr(dy), _ptr(self.w), _ptr(dx), st)
        elif self.wrot is not None:
            L.conv2d_bwd_data_rot(self._dref, _ptr(dy), _ptr(self.w), _ptr(self.wrot), _ptr(add), _ptr(dx), st)
        elif ws is not None and self.bwd_data_ws_bytes:
            L.conv2d_bwd_data_ws(self._dref, _ptr(dy), _ptr(self.w), _ptr(add), _ptr(dx), _ptr(ws), ws.numel() * 4, st)
        else:
            L.conv2d_bwd_data(self._dref, _ptr(dy), _ptr(self.w), _ptr(add), _ptr(dx), st)

    def bwd_data_bn_deferred(self, dy, da, bn: "BatchNorm", bn_x, ld_x, coef, ws, st):
        """bwd_data_bn stopped after its reduction: `da` keeps the plain data gradient, the parameter gradients are written and
        `coef` (3 * cin floats) receives what the apply pass needs - its consumer runs it (ssmi_expand_next_apply)."""
        act = ACT_PRELU if bn.alpha is not None else ACT_NONE
        lib().conv2d_bwd_data_bn_bwd_rot_deferred(self._dref, _ptr(dy), _ptr(self.w), _ptr(self.wrot), _ptr(da), _ptr(bn_x), ld_x,
                                                  _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma), _ptr(bn.beta), act, _ptr(bn.alpha), 0.0,
                                                  _ptr(bn.dgamma), _ptr(bn.dbeta), _ptr(bn.dalpha), _ptr(coef), _ptr(ws), ws.numel() * 4, st)

    def bwd_data_bn(self, dy, da, bn: "BatchNorm", bn_x, ld_x, dx_bn, npix, ws, st, act=None, slope=0.0, drop=None, hw=1):
        """Data gradient `da` (gradient of the output of `bn` + activation, which is this convolution's input) followed by
        the backward of that BatchNormalization: dx_bn = gradient of bn's raw input bn_x (may alias da), parameter
        gradients into the arena.  One fused call for stride-1 convolutions (column sums in the kernel epilogue)."""
        if act is None:
            act = ACT_PRELU if bn.alpha is not None else ACT_NONE
        if self.transpose:
            self.bwd_data(dy, da, st)
            self.store.L.bn_act_drop_bwd(_ptr(da), self.cin, _ptr(bn_x), ld_x, _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma),
                                  _ptr(bn.beta), act, _ptr(bn.alpha), slope, _ptr(drop), hw, _ptr(dx_bn), _ptr(bn.dgamma),
                                  _ptr(bn.dbeta), _ptr(bn.dalpha), npix, self.cin, _ptr(ws), ws.numel() * 4, st)
            return
        self.store.L.conv2d_bwd_data_bn_bwd_rot(self._dref, _ptr(dy), _ptr(self.w), _ptr(self.wrot), None, _ptr(da), _ptr(bn_x),
                                                ld_x, _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma), _ptr(bn.beta), act,
                                                _ptr(bn.alpha), slope, _ptr(drop), hw, _ptr(dx_bn), _ptr(bn.dgamma),
                                                _ptr(bn.dbeta), _ptr(bn.dalpha), _ptr(ws), ws.numel() * 4, st)

    def bwd_data_bn_premerge(self, pm: Premerge, da, bn: "BatchNorm", bn_x, ld_x, dx_bn, ws, st):
        """bwd_data_bn whose output gradient is a merge backward left unformed (`pm`); needs `wrot` and `premerge_ok`."""
        lib().conv2d_bwd_data_bn_bwd_premerge(self._dref, C.byref(pm), _ptr(self.w), _ptr(self.wrot), _ptr(da), _ptr(bn_x), ld_x,
                                              _ptr(bn.mean), _ptr(bn.invstd), _ptr(bn.gamma), _ptr(bn.beta),
                                              ACT_PRELU if bn.alpha is not None else ACT_NONE, _ptr(bn.alpha), 0.0, _ptr(dx_bn),
                                              _ptr(bn.dgamma), _ptr(bn.dbeta), _ptr(bn.dalpha), _ptr(ws), ws.numel() * 4, st)

    def bwd_weight(self, x, dy, ws, st):
        L = self.store.L
        if self.transpose:
            L.conv2d_transpose_bwd_weight(self._dref, _ptr(x), _ptr(dy), _ptr(self.dw), _ptr(self.db), _ptr(ws),
                                          ws.numel() * 4, st)
        else:
            L.conv2d_bwd_weight(self._dref, _ptr(x), _ptr(dy), _ptr(self.dw), _ptr(self.db), _ptr(ws), ws.numel() * 4, st)


class BatchNorm:
    """Keras BatchNormalization(momentum=0.1) optionally fused with the PReLU that follows it."""

    def __init__(self, store: ParamStore, name, c, prelu_name=None):
        self.name, self.c, self.prelu_name = name, c, prelu_name
        store.add(name + "/gamma", (c,), True, "ones")
        store.add(name + "/beta", (c,), True, "zeros")
        store.add(name + "/moving_mean", (c,), False, "zeros")
        store.add(name + "/moving_variance", (c,), False, "ones")
        self.store = store
        self.stats_buf = None

    def add_prelu(self, store, prelu_name):
        self.prelu_name = prelu_name
        store.add(prelu_name + "/alpha", (1, 1, self.c), True, "zeros")

    def plan(self, device):
        v, g = self.store.views, self.store.gviews
        self.gamma, self.beta = v[self.name + "/gamma"], v[self.name + "/beta"]
        self.dgamma, self.dbeta = g[self.name + "/gamma"], g[self.name + "/beta"]
        self.mm, self.mv = v[self.name + "/moving_mean"], v[self.name + "/moving_variance"]
        self.alpha = self.dalpha = None
        if self.prelu_name:
            self.alpha, self.dalpha = v[self.prelu_name + "/alpha"], g[self.prelu_name + "/alpha"]
        sp = self.store.specs
        om, ov = sp[self.name + "/moving_mean"].offset, sp[self.name + "/moving_variance"].offset
        self.mean, self.invstd = self.store.bn_stats[om:om + self.c], self.store.bn_stats[ov:ov + self.c]
        self.params = BnParams(_ptr(self.mean), _ptr(self.invstd), _ptr(self.gamma), _ptr(self.beta))
        self._pref = C.byref(self.params)

    def stats(self, x, npix, ld, training, update_moving, unbiased, ws, st):
        L = self.store.L
        if training:
            L.bn_train_stats(_ptr(x), npix, self.c, ld, BN_EPS, BN_MOMENTUM, 1 if unbiased else 0,
                             _ptr(self.mm) if update_moving else None, _ptr(self.mv) if update_moving else None,
                             _ptr(self.mean), _ptr(self.invstd), _ptr(ws), ws.numel() * 4, st)
        elif not getattr(self.store, "infer_stats_batched", False):     # a graph fills every layer's slice in one launch
            L.bn_infer_stats(_ptr(self.mm), _ptr(self.mv), self.c, BN_EPS, _ptr(self.mean), _ptr(self.invstd), st)

    def act_fwd(self, x, ld, y, npix, st):
        act = ACT_PRELU if self.alpha is not None else ACT_NONE
        self.store.L.bn_act_fwd(_ptr(x), ld, _ptr(self.mean), _ptr(self.invstd), _ptr(self.gamma), _ptr(self.beta), act,
                         _ptr(self.alpha), 0.0, _ptr(y), npix, self.c, st)

    def act_bwd(self, dy, x, ld, dx, npix, ws, st):
        act = ACT_PRELU if self.alpha is not None else ACT_NONE
        self.store.L.bn_act_bwd(_ptr(dy), _ptr(x), ld, _ptr(self.mean), _ptr(self.invstd), _ptr(self.gamma), _ptr(self.beta), act,
                         _ptr(self.alpha), 0.0, _ptr(dx), _ptr(self.dgamma), _ptr(self.dbeta), _ptr(self.dalpha), npix,
                         self.c, _ptr(ws), ws.numel() * 4, st)


def check_queue_budget(side_streams: int, env=None) -> None:
    """A captured step with weight-gradient side streams is a HIP graph with 1 + side_streams concurrent branches.  The
    runtime's graph executor segfaults (host side, at the first launch; exit code 139, nothing printed) when it is asked
    for more concurrent branches than the process has hardware queues: measured with four side streams under
    GPU_MAX_HW_QUEUES=2, and with DEBUG_HIP_FORCE_GRAPH_QUEUES above GPU_MAX_HW_QUEUES.  One side stream (the default)
    runs under every setting tried (GPU_MAX_HW_QUEUES=2, forced graph queues 1-8).  Such a configuration is refused here,
    before anything is captured, instead of crashing later."""
    env = os.environ if env is None else env
    hwq = int(env.get("GPU_MAX_HW_QUEUES", "4"))          # HIP's default is 4 hardware queues per process
    forced = env.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")
    if 1 + side_streams > hwq:
        raise RuntimeError("{} weight-gradient side streams + the main stream need {} hardware queues, GPU_MAX_HW_QUEUES allows {}: "
                           "lower SSMI_WGRAD_STREAMS (1 is the default and the fastest setting) or raise GPU_MAX_HW_QUEUES".format(
                               side_streams, 1 + side_streams, hwq))
    if forced is not None and side_streams > 1 and int(forced) > hwq:
        raise RuntimeError("DEBUG_HIP_FORCE_GRAPH_QUEUES={} exceeds GPU_MAX_HW_QUEUES={} while {} side streams are captured: the "
                           "graph executor crashes in this configuration".format(forced, hwq, side_streams))


_ABLATE_WGRAD = os.environ.get("SSMI_ABLATE_WGRAD") is not None
# timing experiments only (results meaningless): skip the weight gradients of one class of layers - "head" (the logits layer),
# "1x1" (every 1x1 convolution), "spatial" (everything else)
_ABLATE_WGRAD_KIND = os.environ.get("SSMI_ABLATE_WGRAD_KIND", "")
_REDUCE_PER_GROUP = os.environ.get("SSMI_GROUPED_REDUCE", "") == "block"      # grouped strip reductions (opt-in): one launch per block instead of one per join
_WGRAD_LAG = int(os.environ.get("SSMI_WGRAD_LAG", "1"))      # blocks between a fork point and the launches behind it (0: at once)
_WGRAD_FORK_PER_CONV = os.environ.get("SSMI_WGRAD_FORK", "conv") == "conv"      # one fork point per layer instead of one per block


class RunCfg:
    """Per-call switches threaded through the blocks.

    Weight gradients never feed the rest of the backward pass, so they run on a second HIP stream (`side`) with their
    own workspace: `wgrad()` marks the layer's fork point on the main stream (its operands are ready) and collects it,
    `flush()` (called after every block) launches the collected ones on the side stream, each behind its fork point,
    `join()` makes the main stream wait for the side stream.
    Under graph capture this becomes a parallel branch of the HIP graph, which fills the machine while the latency-bound
    main chain (small BN / merge kernels) is running.  Without a side stream wgrad() launches in place."""
    __slots__ = ("training", "update_moving", "unbiased_moving", "use_dropout", "st", "ws", "main", "side", "ws2",
                 "_forked", "_pending", "_sides", "_turn", "_deferred", "defer_reduce", "_reduce_open")

    def __init__(self, training, update_moving, unbiased_moving, use_dropout, st, ws, main=None, side=None, ws2=None):
        self.training, self.update_moving, self.unbiased_moving = training, update_moving, unbiased_moving
        self.use_dropout, self.st, self.ws = use_dropout, st, ws
        # `side` / `ws2`: one stream + workspace, or equally long lists of them (blocks take turns)
        self._sides = None
        if isinstance(side, (list, tuple)):
            self._sides = list(zip(side, ws2))
            side, ws2 = (side[0], ws2[0]) if side else (None, None)
        self.main, self.side, self.ws2 = main, side, ws2
        self._forked = False
        self._pending = []
        self._deferred = []
        self._turn = 0
        # the strip reductions of the weight gradients collected and launched as one kernel at join() (every layer then has its
        # own workspace: conv._ws_view); set by the graph
        self.defer_reduce = False
        self._reduce_open = False

    def wgrad(self, conv, x, dy, pre=None, before=None):
        """`pre`: the BatchNorm (+PReLU) whose output is this layer's input when `x` is that BatchNorm's RAW input.
        `before(stream)`: work that produces `dy` and is needed by nothing else; launched in front of the weight gradient, on
        the stream it runs on."""
        if _ABLATE_WGRAD:       # timing experiment only: the step then computes no weight gradients at all
            return
        if _ABLATE_WGRAD_KIND:
            kind = "head" if conv.name == "logits" else ("1x1" if conv.kh == 1 and conv.kw == 1 else "spatial")
            if kind in _ABLATE_WGRAD_KIND.split(","):
                return
        if self.side is None:
            if before is not None:
                before(self.st)
            if pre is not None:
                conv.bwd_weight_prebn(x, pre, dy, self.ws, self.st)
            else:
                conv.bwd_weight(x, dy, self.ws, self.st)
            return
        ev = None
        if _WGRAD_FORK_PER_CONV:      # the layer's own fork point: its operands are ready now
            ev = torch.cuda.Event()
            ev.record(self.main)
        self._pending.append((conv, x, dy, pre, before, ev))

    def side_job(self, fn, conv=None):
        """`fn(stream)`: work of the weight-gradient chain that is not a weight-gradient launch (the reduction of partial sums a
        main-chain kernel left behind); scheduled like wgrad().  `conv`: the layer it belongs to (ablation switches)."""
        if _ABLATE_WGRAD:
            return
        if _ABLATE_WGRAD_KIND and conv is not None and "1x1" in _ABLATE_WGRAD_KIND.split(","):
            return
        if self.side is None:
            fn(self.st)
            return
        ev = None
        if _WGRAD_FORK_PER_CONV:
            ev = torch.cuda.Event()
            ev.record(self.main)
        self._pending.append((None, None, None, None, fn, ev))

    def flush(self):
        """Puts the weight gradients collected since the last flush on the side stream, each behind its own fork point (or,
        with SSMI_WGRAD_FORK=block, behind everything the main stream has queued so far).  The launches themselves are
        issued at the NEXT flush (or join), after the main stream has queued the kernels of one more block.
        The dependencies are the same either way, but under graph capture the order in which a node's dependents are
        created is the order of its edges, and the HIP runtime keeps a node's FIRST dependent on the node's own hardware
        queue and moves the others to the next queue: launched at once, the side branch would inherit the main chain's queue
        at every fork and the main chain would hop from queue to queue, ending up queued behind weight gradients of
        earlier blocks."""
        if self.side is None:
            return
        self._launch_deferred(max(_WGRAD_LAG - 1, 0))
        if not self._pending:
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self._deferred.append((ev, self._pending))
        self._pending = []
        if _WGRAD_LAG == 0:
            self._launch_deferred(0)

    def _launch_deferred(self, keep):
        while len(self._deferred) > keep:
            self._launch_group(*self._deferred.pop(0))

    def _launch_group(self, ev, items):
        side, ws2 = self.side, self.ws2
        if self._sides:      # blocks take turns (alternating per convolution instead measured slower)
            side, ws2 = self._sides[self._turn % len(self._sides)]
            self._turn += 1
        if not _WGRAD_FORK_PER_CONV:
            side.wait_event(ev)
        if self.defer_reduce and not self._reduce_open:
            lib().wg_reduce_defer_begin()
            self._reduce_open = True
        for conv, x, dy, pre, before, own in items:
            if own is not None:
                side.wait_event(own)
            if before is not None:
                before(side.cuda_stream)
            if conv is None:      # side_job
                continue
            wsc = conv._ws_view if (self.defer_reduce and conv._ws_view is not None) else ws2
            if pre is not None:
                conv.bwd_weight_prebn(x, pre, dy, wsc, side.cuda_stream)
            else:
                conv.bwd_weight(x, dy, wsc, side.cuda_stream)
        if self._reduce_open and _REDUCE_PER_GROUP:
            lib().wg_reduce_defer_flush(0, side.cuda_stream)
        self._forked = True

    def join(self):
        self.flush()
        self._launch_deferred(0)
        if self._reduce_open:
            lib().wg_reduce_defer_flush(1, self.side.cuda_stream)
            self._reduce_open = False
        if self.side is not None and self._forked:
            for side in ([sd for sd, _ in self._sides] if self._sides else [self.side]):
                ev = torch.cuda.Event()
                ev.record(side)
                self.main.wait_event(ev)
            self._forked = False


# ----------------------------------------------------------------------------------------------------------
# fused blocks
# ----------------------------------------------------------------------------------------------------------

class Block:
    name = ""

    def convs(self) -> List[Conv]:
        return []

    def dropout(self) -> Optional[Tuple[str, float, int]]:
        return None

    def dropouts(self) -> List[Tuple[str, float, int]]:
        d = self.dropout()
        return [] if d is None else [d]

    def bind_dropout(self, views: Dict[str, torch.Tensor]):
        d = self.dropout()
        if d is not None:
            self.drop_mask = views[d[0]]


class InitialBlock(Block):
    """conv3x3/s2 (3->13) || maxpool2x2 (3) -> concat(16) [-> BN -> PReLU in the naive variant]
    (src/models.py:1055-1066 / 1361-1370).  The concat is never materialised separately: both producers write
    channel slices of one 16-channel buffer."""

    def __init__(self, store, variant):
        self.name = "initial_block"
        self.store = store
        self.variant = variant
        self.conv = Conv(store, "initial_block_conv2d", 3, 3, 3, 13, stride=2, padding="same", bias=True,
                         init="glorot_uniform")
        self.bn = None
        if variant in ("naive", "enhanced"):
            self.bn = BatchNorm(store, "initial_block_bnorm", 16)
            self.bn.add_prelu(store, "initial_block_prelu")

    def convs(self):
        return [self.conv]

    def plan(self, n, h, w, dev):
        self.n, self.h, self.w = n, h, w
        self.ho, self.wo = self.conv.plan(n, h, w, x_ld=3, y_ld=16)
        assert h % 2 == 0 and w % 2 == 0, "input height/width must be even"
        self.npix = n * self.ho * self.wo
        self.cat = torch.empty(n, self.ho, self.wo, 16, dtype=self.store.act_dtype, device=dev)
        self.pool_slice = self.cat.view(-1)[13:]
        self.idx = torch.empty(n, self.ho, self.wo, 3, dtype=torch.int32, device=dev) if self.variant == "unpool" else None
        if self.bn:
            self.bn.plan(dev)
            self.out = torch.empty_like(self.cat)
            self.dcat = torch.empty_like(self.cat)
        else:
            self.out = self.cat
        return (self.ho, self.wo, 16)

    def fwd(self, x, cfg: RunCfg):
        L = self.store.L
        self.x = x
        self.conv.fwd(x, self.cat, cfg.st)
        L.maxpool_argmax_fwd(_ptr(x), self.n, self.h, self.w, 3, _ptr(self.pool_slice), 16, _ptr(self.idx), None, cfg.st)
        if self.bn:
            self.bn.stats(self.cat, self.npix, 16, cfg.training, cfg.update_moving, cfg.unbiased_moving, cfg.ws, cfg.st)
            self.bn.act_fwd(self.cat, 16, self.out, self.npix, cfg.st)
        return self.out

    def bwd(self, dout, cfg: RunCfg):
        if self.bn:
            self.bn.act_bwd(dout, self.cat, 16, self.dcat, self.npix, cfg.ws, cfg.st)
            d = self.dcat
        else:
            d = dout
        cfg.wgrad(self.conv, self.x, d)
        return None  # the images need no gradient


_NO_FUSED_INFER_MERGE = os.environ.get("SSMI_NO_FUSED_INFER_MERGE") is not None    # A/B switch for measurements


class DeferredInputGradient:
    """The input gradient of a regular bottleneck, not formed yet: dx = dy1 * W1^T + dsum (the data gradient of its 1x1
    reduction, src/models.py:1090-1096, plus its skip gradient).  A bottleneck in front that recomputes its expansion consumes
    the three operands in its own merge backward (ssmi_expand_merge_bwd_chained: dx is neither written nor read); any other
    consumer gets the tensor from materialise()."""

    def __init__(self, block, apply_pending=False):
        self.block = block
        # the backward of the block's BN1 + PReLU1 stopped after its reduction (block.coef1): block.da1 still holds the gradient
        # BEHIND them, and the consumer's reducing pass applies them (ssmi_expand_next_apply) and writes dy1 back to block.da1
        self.apply_pending = apply_pending

    def weight_gradient(self, cfg):
        """The reduction's weight gradient as its own launch (a consumer that forms it itself does not call this)."""
        b = self.block
        cfg.wgrad(b.conv1, b.x, b.da1)

    def materialise(self, cfg, weight_gradient=True):
        b = self.block
        if self.apply_pending:
            raise RuntimeError("the input gradient of {} was deferred together with its BN1 backward's apply pass, which only the "
                               "chained merge backward of the block in front can run".format(b.name))
        if weight_gradient:
            self.weight_gradient(cfg)
        b.conv1.bwd_data(b.da1, b.dx, cfg.st, add=b.dsum)
        return b.dx


def run_backward(blocks_last_to_first, d, cfg, last=True):
    """Backward pass over blocks (given last block first), weight gradients flushed to the side chain after every block.
    Returns the gradient of the first block's input: a tensor, or - `last` False: more blocks follow in another call, which takes
    what this one returns as its `d` - possibly a DeferredInputGradient."""
    for b in blocks_last_to_first:
        if isinstance(d, DeferredInputGradient) and not (isinstance(b, Bottleneck) and b.accepts(d)):
            d = d.materialise(cfg)
        d = b.bwd(d, cfg)
        cfg.flush()
    return d.materialise(cfg) if (last and isinstance(d, DeferredInputGradient)) else d


class Bottleneck(Block):
    """Encoder bottleneck (src/models.py:1069-1174 / 1373-1481) and decoder bottleneck (1208-1299 / 1520-1610).

    main : conv1 (1x1, or 2x2/s2 when downsampling) -> BN1+PReLU1 -> conv2 (3x3 | dilated 3x3 | 1x5 then 5x1 |
           transposed 3x3/s2) -> BN2+PReLU2 -> conv3 (1x1) -> BN3 [-> SpatialDropout2D]
    other: identity | maxpool2x2 + channel zero-pad | 1x1 conv -> BN -> (nearest upsample | max-unpool)
    out  : PReLU3(main + other), computed by one fused merge kernel.
    """

    def __init__(self, store, prefix, cin, cout, kind, variant, downsample=False, upsample=False, asymmetric=0, dilated=0,
                 dropout_rate=None):
        self.name = prefix
        self.store = store
        self.cin, self.cout, self.kind, self.variant = cin, cout, kind, variant
        self.downsample, self.upsample = downsample, upsample
        internal = cout // 4
        self.internal = internal
        s = 2 if downsample else 1
        self.conv1 = Conv(store, prefix + "_proj_conv2d_1", s, s, cin, internal, stride=s, padding="valid", bias=False)
        self.bn1 = BatchNorm(store, prefix + "_bnorm_1", internal)
        self.bn1.add_prelu(store, prefix + "_prelu_1")
        self.conv2a = None
        self.nn_resize = upsample and variant == "enhanced"
        if self.nn_resize:
            # NN-resize convolution of the enhanced decoder (src/models.py:1711-1719): nearest 2x upsampling + 3x3 conv
            self.conv2 = Conv(store, prefix + "_nn_rs_conv_conv2d", 3, 3, internal, internal, bias=True)
        elif upsample:
            self.conv2 = Conv(store, prefix + "_tconv2d_1", 3, 3, internal, internal, stride=2, bias=True, transpose=True)
        elif asymmetric:
            self.conv2a = Conv(store, prefix + "_aconv2d_1", 1, asymmetric, internal, internal, bias=False)
            self.conv2 = Conv(store, prefix + "_aconv2d_2", asymmetric, 1, internal, internal, bias=True)
        elif dilated:
            self.conv2 = Conv(store, prefix + "_dconv2d", 3, 3, internal, internal, dil=dilated, bias=True)
        else:
            self.conv2 = Conv(store, prefix + "_conv2d_1", 3, 3, internal, internal, bias=True)
        self.bn2 = BatchNorm(store, prefix + "_bnorm_2", internal)
        self.bn2.add_prelu(store, prefix + "_prelu_2")
        self.conv3 = Conv(store, prefix + "_proj_conv2d_2", 1, 1, internal, cout, padding="valid", bias=False)
        self.conv_o = self.bn_o = None
        has_other_conv = kind == "dec" and (cin != cout or upsample)
        if has_other_conv and not upsample:
            raise NotImplementedError("decoder bottleneck with a projected, non-upsampled skip branch")

        def make_other():
            self.conv_o = Conv(store, prefix + "_other_conv2d", 1, 1, cin, cout, padding="valid", bias=False)
            self.bn_o = BatchNorm(store, prefix + "_other_bnorm_1", cout)

        # weight order follows layer creation order in the reference builders
        if kind == "dec" and variant == "unpool":
            if has_other_conv:
                make_other()
            self.bn3 = BatchNorm(store, prefix + "_bnorm_3", cout)
        else:
            self.bn3 = BatchNorm(store, prefix + "_bnorm_3", cout)
            if has_other_conv:
                make_other()
        store.add(prefix + "_prelu_3/alpha", (1, 1, cout), True, "zeros")
        self.drop_rate = dropout_rate   # None: no SpatialDropout2D layer in this block
        self.drop_mask = None           # [n, cout] view set by the network
        if downsample:
            self.mode = 1
        elif upsample:
            self.mode = 3 if variant == "unpool" else 2
        else:
            self.mode = 0

    def convs(self):
        return [c for c in (self.conv1, self.conv2a, self.conv2, self.conv3, self.conv_o) if c is not None]

    def dropout(self):
        if self.drop_rate is None:
            return None
        return (self.name + "_sdrop2d_1", self.drop_rate, self.cout)

    def plan(self, n, h, w, dev, unpool_idx=None):
        """(h, w): input spatial size.  unpool_idx: int32 indices for mode 3."""
        self.n = n
        f32 = dict(dtype=self.store.act_dtype, device=dev)
        h1, w1 = self.conv1.plan(n, h, w)
        h2, w2 = h1, w1
        if self.conv2a is not None:
            self.conv2a.plan(n, h1, w1)
            self.t = torch.empty(n, h1, w1, self.internal, **f32)
            self.dt = torch.empty_like(self.t)
        if self.nn_resize:
            self.a1u = torch.empty(n, 2 * h1, 2 * w1, self.internal, **f32)
            self.da1u = torch.empty_like(self.a1u)
            h2, w2 = self.conv2.plan(n, 2 * h1, 2 * w1)
        else:
            h2, w2 = self.conv2.plan(n, h1, w1)
        self.conv3.plan(n, h2, w2)
        self.hin, self.win, self.h1, self.w1, self.ho, self.wo = h, w, h1, w1, h2, w2
        self.np_in, self.np1, self.np_out = n * h * w, n * h1 * w1, n * h2 * w2
        i, c = self.internal, self.cout
        self.y1 = torch.empty(n, h1, w1, i, **f32)
        self.a1 = torch.empty_like(self.y1)
        self.y2 = torch.empty(n, h2, w2, i, **f32)
        self.a2 = torch.empty_like(self.y2)
        self.y3 = torch.empty(n, h2, w2, c, **f32)
        self.out = torch.empty_like(self.y3)
        # gradients
        self.dy3 = torch.empty_like(self.y3)
        self.dsum = torch.empty_like(self.y3)
        self.coef3 = torch.empty(3 * c, dtype=torch.float32, device=dev)     # merge backward's coefficients (deferred apply pass)
        self.da2 = torch.empty_like(self.a2)   # reused in place for dy2
        self.da1 = torch.empty_like(self.a1)   # reused in place for dy1
        self.dx = torch.empty(n, h, w, self.cin, **f32)
        for b in (self.bn1, self.bn2, self.bn3):
            b.plan(dev)
        self.alpha3 = self.bn3.store.views[self.name + "_prelu_3/alpha"]
        self.dalpha3 = self.bn3.store.gviews[self.name + "_prelu_3/alpha"]
        self.pooled = self.idx = self.yo = None
        if self.mode == 1:
            self.pooled = torch.empty(n, h2, w2, self.cin, **f32)
            self.idx = torch.empty(n, h2, w2, self.cin, dtype=torch.int32, device=dev)
        if self.mode >= 2:
            self.conv_o.plan(n, h, w)
            self.bn_o.plan(dev)
            self.yo = torch.empty(n, h, w, c, **f32)
            self.dyo_bn = torch.empty_like(self.yo)
            self.dx_main = torch.empty_like(self.dx)
            self.unpool_idx = unpool_idx
            if self.mode == 3:
                assert unpool_idx is not None and tuple(unpool_idx.shape) == (n, h, w, c), \
                    "max-unpooling indices {} do not match the skip tensor {}".format(
                        None if unpool_idx is None else tuple(unpool_idx.shape), (n, h, w, c))
        md = MergeDesc()
        md.n, md.h, md.w, md.c, md.other_mode, md.c_other = n, h2, w2, c, self.mode, (self.cin if self.mode == 1 else c)
        md.dtype = self.store.dtype_code
        self.md = md
        self._mref = C.byref(md)
        # Regular and downsampling bottlenecks, training phase: the expansion's output y3 is never stored - statistics, merge
        # forward and both merge-backward passes recompute it from the raw output of conv2 (ssmi_expand_*).  SSMI_NO_EXPAND=1
        # keeps the stored route; keep_y3 (tests) has the forward kernel store y3 as well.
        self.expand = None
        # (both storage types; with bfloat16 storage the products run on the bfloat16 matrix instructions)
        if self.mode <= 1 and not os.environ.get("SSMI_NO_EXPAND") and not self.conv3.premerge_ok:
            ed = ExpandDesc()
            ed.n, ed.h, ed.w, ed.cin, ed.cout, ed.dtype = n, h2, w2, i, c, self.store.dtype_code
            if lib().expand_supported(C.byref(ed)):
                self.expand = ed
                self._eref = C.byref(ed)
        self._expanded = False
        self._a1_ready = False
        self.infer_expand = not os.environ.get("SSMI_NO_INFER_EXPAND")
        self.prev_block = None
        last2 = self.conv2a if self.conv2a is not None else self.conv2      # the layer whose data gradient meets BN1
        # (on for bfloat16 storage only: 5.67 against 5.72 ms per ENet step; float32 7.94 against 7.87 - the chained kernel gets
        # heavier and the reduction's weight gradient can no longer start beside it.  SSMI_DEFER_APPLY=1 / 0 overrides.)
        want = os.environ.get("SSMI_DEFER_APPLY", "0" if self.store.dtype_code == F32 else "1") == "1"
        if self.store.dtype_code == F32 and os.environ.get("SSMI_FUSED_WGRAD", "0") == "1":
            want = False      # (float32: the chained kernel has no registers for the fused weight gradient AND the apply pass)
        self.defer_apply_ok = (want and not self.nn_resize and not last2.transpose and
                               bool(lib().conv2d_bwd_data_bn_bwd_deferred_supported(last2._dref, i)))
        self.coef1 = torch.empty(3 * i, dtype=torch.float32, device=dev)
        na = ExpandNextApply()
        na.y1, na.bn1, na.alpha, na.coef = _ptr(self.y1), C.pointer(self.bn1.params), _ptr(self.bn1.alpha), _ptr(self.coef1)
        self._next_apply = na
        self._y1_ready = False
        self.keep_y3 = bool(os.environ.get("SSMI_KEEP_Y3"))
        # A regular bottleneck whose own expansion is recomputed has the shapes the chained merge backward of the block in front
        # needs (cin -> cin/4 reduction): it hands its input gradient over unformed (DeferredInputGradient).  SSMI_NO_CHAIN=1: never.
        # The expansion's data gradient + BN2 backward sums inside the merge backward's apply pass (ssmi_expand_merge_bwd_through):
        # on for the 64-channel blocks.  Measured on the ENet 512x512 batch-16 step (ms): neither 8.06, 64 only 8.01, 128 only
        # 8.20, both 8.13 - for 128 channels the main chain alone gets 0.15 ms shorter but the expansion's weight gradient no
        # longer reads dy3 alongside the data gradient it used to overlap with, and the step gets longer.
        # (bfloat16 storage, where the extra product is nearly free: both on, 5.93 against 5.98 ms)
        self.through = (not os.environ.get("SSMI_NO_THROUGH") and
                        str(c) in os.environ.get("SSMI_THROUGH_C", "64" if self.store.dtype_code == F32 else "64,128").split(","))
        # SSMI_FUSED_WGRAD=1: the weight gradients of the two 1x1 layers around a chained / through merge backward are formed inside
        # it (per-workgroup partial matrices in wpart3 / wpart1, summed on the weight-gradient chain) instead of by their own
        # launches.  Off by default: it removes 0.8 GB and ~45 launches from the weight-gradient chain but lengthens the main
        # chain's kernels, and the main chain is the critical path - ENet step 8.11 against 8.00 ms (float32), 5.93 against 5.85
        # (bfloat16).
        self.fuse_wg = self.expand is not None and os.environ.get("SSMI_FUSED_WGRAD", "0") == "1"
        self.wpart3 = self.wpart1 = None
        if self.fuse_wg:
            nfl = lib().expand_weight_partial_bytes(self._eref) // 4
            if self.through:
                self.wpart3 = torch.empty(nfl, dtype=torch.float32, device=dev)
            if self.mode == 0 and self.cin == self.cout:
                self.wpart1 = torch.empty(nfl, dtype=torch.float32, device=dev)      # filled by the block in front (chained)
        self.defers = self.mode == 0 and self.expand is not None and self.cin == self.cout and not os.environ.get("SSMI_NO_CHAIN")
        return (h2, w2, c)

    def accepts(self, d: DeferredInputGradient):
        nb = d.block
        return (self._expanded and self.cout == nb.cin and self.expand.cin == nb.internal
                and (self.n, self.ho, self.wo) == (nb.n, nb.hin, nb.win))

    def _merge_other(self):
        if self.mode == 0:
            return self.x, None, None
        if self.mode == 1:
            return self.pooled, None, None
        return self.yo, self.bn_o._pref, (self.unpool_idx if self.mode == 3 else None)

    def fwd(self, x, cfg: RunCfg):
        L = self.store.L
        st, ws = cfg.st, cfg.ws
        tr, um, ub = cfg.training, cfg.update_moving, cfg.unbiased_moving
        self.x = x
        i = self.internal
        # inference phase: BN (fixed statistics) + PReLU ride in the convolution's epilogue, a1 / a2 are written directly
        fuse1 = not tr and not self.conv1.transpose
        fuse2 = not tr and not self.conv2.transpose
        if not tr and self._a1_ready:      # inference phase: the block in front has left the activated a1 (merge forward carried through)
            self._a1_ready = False
        elif fuse1:
            self.conv1.fwd_bn_act_infer(x, self.a1, self.bn1, st)
        else:
            if tr and self._y1_ready:      # the block in front has left y1 and BN1's statistics (merge forward carried through)
                self._y1_ready = False
            else:
                self.conv1.fwd_bn(x, self.y1, self.bn1, self.np1, tr, um, ub, ws, st)
            self.bn1.act_fwd(self.y1, i, self.a1, self.np1, st)
        src2 = self.a1
        if self.conv2a is not None:
            self.conv2a.fwd(self.a1, self.t, st)
            src2 = self.t
        elif self.nn_resize:
            L.upsample_nn2x_fwd(_ptr(self.a1), self.n, self.h1, self.w1, i, _ptr(self.a1u), i, st)
            src2 = self.a1u
        # inference phase, expansion recomputed inside the merge (and carried through the next block's reduction): conv2 stores
        # its RAW output, its BN + PReLU are applied on load like in the training phase.  SSMI_NO_INFER_EXPAND=1: the route below.
        infer_expand = (not tr and self.expand is not None and self.infer_expand and not self.conv2.transpose
                        and not self.keep_y3)
        if infer_expand:
            self.conv2.fwd(src2, self.y2, st)
            drop = self.drop_mask if (cfg.use_dropout and self.drop_mask is not None) else None
            self._drop_used = drop
            self._expanded = False
            bn2, bn3, L0 = self.bn2, self.bn3, lib()
            skip, c_skip = x, self.cout
            if self.mode == 1:
                L.maxpool_argmax_fwd(_ptr(x), self.n, self.hin, self.win, self.cin, _ptr(self.pooled), self.cin, _ptr(self.idx), None, st)
                skip, c_skip = self.pooled, self.cin
            nb = getattr(self, "next_fwd", None)
            if nb is not None:
                L0.expand_merge_fwd_through_infer(self._eref, _ptr(self.y2), bn2._pref, _ptr(bn2.alpha), _ptr(self.conv3.w), bn3._pref,
                                                  _ptr(drop), _ptr(skip), c_skip, _ptr(self.alpha3), _ptr(self.out), _ptr(nb.conv1.w),
                                                  nb.bn1._pref, _ptr(nb.bn1.alpha), _ptr(nb.a1), st)
                nb._a1_ready = True
            else:
                L0.expand_merge_fwd(self._eref, _ptr(self.y2), bn2._pref, _ptr(bn2.alpha), _ptr(self.conv3.w), bn3._pref, _ptr(drop),
                                    _ptr(skip), c_skip, _ptr(self.alpha3), _ptr(self.out), None, st)
            return self.out
        if fuse2:
            self.conv2.fwd_bn_act_infer(src2, self.a2, self.bn2, st)
        else:
            self.conv2.fwd_bn(src2, self.y2, self.bn2, self.np_out, tr, um, ub, ws, st)
            # training phase: BN2 + PReLU2 are applied by conv3 (and by its weight gradient) to their operand; a2 is not stored
            self._pre3 = tr and self.conv3.prebn_ok
            if not self._pre3:
                self.bn2.act_fwd(self.y2, i, self.a2, self.np_out, st)
        if (not tr and self.mode == 0 and not (cfg.use_dropout and self.drop_mask is not None)
                and not _NO_FUSED_INFER_MERGE):
            # inference phase, identity skip: BN3, the Add and PReLU3 ride in conv3's epilogue (dropout is the identity)
            self._drop_used = None
            self.conv3.fwd_bn_add_act_infer(self.a2, self.out, self.bn3, x, self.cin, self.alpha3, st)
            return self.out
        self._expanded = bool(tr and getattr(self, "_pre3", False) and self.expand is not None)
        if self._expanded:
            drop = self.drop_mask if (cfg.use_dropout and self.drop_mask is not None) else None
            self._drop_used = drop
            bn2, bn3, L0 = self.bn2, self.bn3, lib()
            L0.expand_bn_stats(self._eref, _ptr(self.y2), bn2._pref, _ptr(bn2.alpha), _ptr(self.conv3.w), BN_EPS, BN_MOMENTUM,
                               1 if ub else 0, _ptr(bn3.mm) if um else None, _ptr(bn3.mv) if um else None, _ptr(bn3.mean),
                               _ptr(bn3.invstd), _ptr(ws), ws.numel() * 4, st)
            skip, c_skip = x, self.cout
            if self.mode == 1:      # downsampling block: the skip branch is the max-pooled input, channel-padded with zeros
                L.maxpool_argmax_fwd(_ptr(x), self.n, self.hin, self.win, self.cin, _ptr(self.pooled), self.cin, _ptr(self.idx), None, st)
                skip, c_skip = self.pooled, self.cin
            nb = getattr(self, "next_fwd", None)
            if nb is not None:
                b1 = nb.bn1
                L0.expand_merge_fwd_through(self._eref, _ptr(self.y2), bn2._pref, _ptr(bn2.alpha), _ptr(self.conv3.w), bn3._pref, _ptr(drop),
                                            _ptr(skip), c_skip, _ptr(self.alpha3), _ptr(self.out), _ptr(self.y3) if self.keep_y3 else None,
                                            _ptr(nb.conv1.w), _ptr(nb.y1), BN_EPS, BN_MOMENTUM, 1 if ub else 0,
                                            _ptr(b1.mm) if um else None, _ptr(b1.mv) if um else None, _ptr(b1.mean), _ptr(b1.invstd),
                                            _ptr(ws), ws.numel() * 4, st)
                nb._y1_ready = True
                return self.out
            L0.expand_merge_fwd(self._eref, _ptr(self.y2), bn2._pref, _ptr(bn2.alpha), _ptr(self.conv3.w), bn3._pref, _ptr(drop),
                                _ptr(skip), c_skip, _ptr(self.alpha3), _ptr(self.out), _ptr(self.y3) if self.keep_y3 else None, st)
            return self.out
        if tr and getattr(self, "_pre3", False):
            self.conv3.fwd_bn_prebn(self.y2, self.bn2, self.y3, self.bn3, um, ub, ws, st)
        else:
            self.conv3.fwd_bn(self.a2, self.y3, self.bn3, self.np_out, tr, um, ub, ws, st)
        if self.mode == 1:
            L.maxpool_argmax_fwd(_ptr(x), self.n, self.hin, self.win, self.cin, _ptr(self.pooled), self.cin,
                                 _ptr(self.idx), None, st)
        elif self.mode >= 2:
            self.conv_o.fwd_bn(x, self.yo, self.bn_o, self.np_in, tr, um, ub, ws, st)
        other, bn_o, idx = self._merge_other()
        drop = self.drop_mask if (cfg.use_dropout and self.drop_mask is not None) else None
        self._drop_used = drop
        L.merge_fwd(self._mref, _ptr(self.y3), self.bn3._pref, _ptr(drop), _ptr(other), bn_o, _ptr(idx), _ptr(self.alpha3),
                    _ptr(self.out), st)
        return self.out

    def bwd(self, dout, cfg: RunCfg):
        L = self.store.L
        st, ws = cfg.st, cfg.ws
        wsb = ws.numel() * 4
        i = self.internal
        other, bn_o, idx = self._merge_other()
        # The merge backward's last pass (dy3 from dsum, y3 and three coefficient vectors) is off the main chain where conv3's
        # data gradient can apply it to its own operand: dy3 is then stored only for conv3's weight gradient, in front of it
        # on its stream.
        premerge = self.conv3.premerge_ok and self.conv3.wrot is not None
        before3 = None
        self._chained = self._expanded and isinstance(dout, DeferredInputGradient)
        # (SSMI_NO_THROUGH=1: the expansion's data gradient as its own launch)
        through = self._expanded and self.through
        if self._expanded:
            nb = dout.block if self._chained else None
            head = (self._eref,) + ((None, _ptr(nb.da1), _ptr(nb.conv1.w), _ptr(nb.dsum)) if self._chained else (_ptr(dout), None, None, None))
            body = (_ptr(self.y2), self.bn2._pref, _ptr(self.bn2.alpha), _ptr(self.conv3.w), self.bn3._pref, _ptr(self._drop_used),
                    _ptr(self.x if self.mode == 0 else self.pooled), self.cout if self.mode == 0 else self.cin, _ptr(self.alpha3),
                    _ptr(self.dy3), _ptr(self.dsum), _ptr(self.bn3.dgamma), _ptr(self.bn3.dbeta), _ptr(self.dalpha3))
            # fused weight gradients: conv3's own rides on the through route, the next block's conv1's on the chained route
            # (which then has to take the through entry point as well - only it carries the partial buffers)
            wp3 = self.wpart3 if through else None
            wp1 = nb.wpart1 if self._chained else None
            na = None
            if self._chained and dout.apply_pending:
                na = nb._next_apply
            # the next block's reduction's weight gradient: forked in FRONT of the launch below where it can be (it then runs
            # beside it: 7.87 against 8.05 ms per float32 ENet step); with a pending apply pass that launch is what writes its dy1
            if self._chained and wp1 is None and na is None:
                dout.weight_gradient(cfg)
            if through or wp1 is not None or na is not None:
                bn2 = self.bn2
                if wp3 is not None:
                    body = body[:9] + (None,) + body[10:]      # dmain is not stored
                thru = (_ptr(self.da2), _ptr(bn2.dgamma), _ptr(bn2.dbeta), _ptr(bn2.dalpha)) if through else (None, None, None, None)
                lib().expand_merge_bwd_through(*head, *body, *thru, _ptr(wp3), _ptr(wp1), C.byref(na) if na is not None else None,
                                               _ptr(ws), wsb, st)
            elif self._chained:
                lib().expand_merge_bwd_chained(head[0], *head[2:], *body, _ptr(ws), wsb, st)
            else:
                lib().expand_merge_bwd(head[0], head[1], *body, _ptr(ws), wsb, st)
            if self._chained and wp1 is None and na is not None:
                dout.weight_gradient(cfg)
            if wp1 is not None:      # behind the launch that left the partial sums
                cfg.side_job(lambda stream, e=nb._eref, wp=wp1, dw=nb.conv1.dw: lib().expand_weight_reduce(e, _ptr(wp), _ptr(dw), stream),
                             conv=nb.conv1)
        elif premerge:
            L.merge_bwd_deferred(self._mref, _ptr(dout), _ptr(self.y3), self.bn3._pref, _ptr(self._drop_used), _ptr(other), bn_o,
                                 _ptr(idx), _ptr(self.alpha3), _ptr(self.dsum), _ptr(self.dyo_bn if self.mode >= 2 else None),
                                 _ptr(self.bn3.dgamma), _ptr(self.bn3.dbeta), _ptr(self.dalpha3), _ptr(self.coef3), _ptr(ws), wsb, st)
            drop3 = self._drop_used

            def before3(stream, L=L, drop3=drop3):
                L.merge_bwd_apply(self._mref, _ptr(self.dsum), _ptr(self.y3), self.bn3._pref, _ptr(drop3), _ptr(self.coef3),
                                  _ptr(self.dy3), stream)
        else:
            L.merge_bwd(self._mref, _ptr(dout), _ptr(self.y3), self.bn3._pref, _ptr(self._drop_used), _ptr(other), bn_o, _ptr(idx),
                        _ptr(self.alpha3), _ptr(self.dy3), _ptr(self.dsum), _ptr(self.dyo_bn if self.mode >= 2 else None),
                        _ptr(self.bn3.dgamma), _ptr(self.bn3.dbeta), _ptr(self.dalpha3), _ptr(ws), wsb, st)
        # conv3
        if self._expanded and through and self.wpart3 is not None:
            cfg.side_job(lambda stream, e=self._eref, wp=self.wpart3, dw=self.conv3.dw: lib().expand_weight_reduce(e, _ptr(wp), _ptr(dw), stream),
                         conv=self.conv3)
        elif getattr(self, "_pre3", False):
            cfg.wgrad(self.conv3, self.y2, self.dy3, pre=self.bn2, before=before3)
        else:
            cfg.wgrad(self.conv3, self.a2, self.dy3, before=before3)
        if through:
            pass      # da2 -> dy2 came out of the merge backward
        elif premerge:
            pm = Premerge(_ptr(self.dsum), _ptr(self.y3), _ptr(self.bn3.mean), _ptr(self.bn3.invstd), _ptr(self.coef3),
                          _ptr(self._drop_used), self.ho * self.wo)
            self.conv3.bwd_data_bn_premerge(pm, self.da2, self.bn2, self.y2, i, self.da2, ws, st)      # da2 -> dy2 (in place)
        else:
            self.conv3.bwd_data_bn(self.dy3, self.da2, self.bn2, self.y2, i, self.da2, self.np_out, ws, st)   # da2 -> dy2 (in place)
        # conv2.  A block whose input gradient the block in front will certainly form itself (chained merge backward) leaves the
        # apply pass of its BN1 backward to that kernel as well (bfloat16 storage by default, see plan())
        pb = self.prev_block
        defer_apply = bool(self.mode == 0 and self.defers and self._expanded and self.defer_apply_ok and pb is not None
                           and pb._expanded and pb.accepts(DeferredInputGradient(self)))
        if self.conv2a is not None:
            cfg.wgrad(self.conv2, self.t, self.da2)
            self.conv2.bwd_data(self.da2, self.dt, st)
            cfg.wgrad(self.conv2a, self.a1, self.dt)
            if defer_apply:
                self.conv2a.bwd_data_bn_deferred(self.dt, self.da1, self.bn1, self.y1, i, self.coef1, ws, st)
            else:
                self.conv2a.bwd_data_bn(self.dt, self.da1, self.bn1, self.y1, i, self.da1, self.np1, ws, st)
        elif self.nn_resize:
            cfg.wgrad(self.conv2, self.a1u, self.da2)
            self.conv2.bwd_data(self.da2, self.da1u, st)
            L.upsample_nn2x_bwd(_ptr(self.da1u), i, self.n, self.h1, self.w1, i, _ptr(self.da1), st)
            self.bn1.act_bwd(self.da1, self.y1, i, self.da1, self.np1, ws, st)     # da1 -> dy1 (in place)
        else:
            cfg.wgrad(self.conv2, self.a1, self.da2)
            if defer_apply:
                self.conv2.bwd_data_bn_deferred(self.da2, self.da1, self.bn1, self.y1, i, self.coef1, ws, st)
            else:
                self.conv2.bwd_data_bn(self.da2, self.da1, self.bn1, self.y1, i, self.da1, self.np1, ws, st)   # da1 -> dy1
        # conv1 + skip branch
        if self.mode == 0 and self.defers and self._expanded:
            return DeferredInputGradient(self, defer_apply)      # (conv1's weight gradient: by whoever forms the input gradient)
        cfg.wgrad(self.conv1, self.x, self.da1)
        if self.mode == 0:
            self.conv1.bwd_data(self.da1, self.dx, st, add=self.dsum)
        elif self.mode == 1:
            self.conv1.bwd_data(self.da1, self.dx, st)
            L.maxpool_argmax_bwd(_ptr(self.dsum), self.cout, _ptr(self.idx), self.n, self.hin, self.win, self.cin,
                                 _ptr(self.dx), 1, st)
        else:
            self.conv1.bwd_data(self.da1, self.dx_main, st)
            self.bn_o.act_bwd(self.dyo_bn, self.yo, self.cout, self.dyo_bn, self.np_in, ws, st)   # -> d(yo) in place
            cfg.wgrad(self.conv_o, self.x, self.dyo_bn)
            self.conv_o.bwd_data(self.dyo_bn, self.dx, st, add=self.dx_main)
        return self.dx


class EnhancedHead(Block):
    """Head of the enhanced decoder (src/models.py:1779-1794): nearest 2x upsampling -> 3x3 conv (16 -> classes) ->
    BN -> PReLU -> 1x1 conv named 'logits'."""

    def __init__(self, store, cin, num_classes, final_layer_name="logits"):
        self.name = final_layer_name
        self.store = store
        self.cin, self.num_classes = cin, num_classes
        self.conv_a = Conv(store, "final_nn_rs_conv_conv2d", 3, 3, cin, num_classes, bias=True)
        self.bn = BatchNorm(store, "final_nn_rs_conv_bnorm", num_classes)
        self.bn.add_prelu(store, "final_nn_rs_conv_prelu")
        self.conv = Conv(store, final_layer_name, 1, 1, num_classes, num_classes, padding="valid", bias=True)

    def convs(self):
        return [self.conv_a, self.conv]

    def plan(self, n, h, w, dev):
        f32 = dict(dtype=self.store.act_dtype, device=dev)
        self.n, self.h, self.w = n, h, w
        H, W, C_ = 2 * h, 2 * w, self.num_classes
        self.conv_a.plan(n, H, W)
        self.conv.plan(n, H, W)
        self.bn.plan(dev)
        self.npix = n * H * W
        self.xu = torch.empty(n, H, W, self.cin, **f32)
        self.ya = torch.empty(n, H, W, C_, **f32)
        self.aa = torch.empty_like(self.ya)
        self.out = torch.empty_like(self.ya)
        self.dout = torch.empty_like(self.ya)
        self.daa = torch.empty_like(self.ya)
        self.dya = torch.empty_like(self.ya)
        self.dxu = torch.empty_like(self.xu)
        self.dx = torch.empty(n, h, w, self.cin, **f32)
        return (H, W, C_)

    def fwd(self, x, cfg):
        L = self.store.L
        C_ = self.num_classes
        L.upsample_nn2x_fwd(_ptr(x), self.n, self.h, self.w, self.cin, _ptr(self.xu), self.cin, cfg.st)
        self.conv_a.fwd_bn(self.xu, self.ya, self.bn, self.npix, cfg.training, cfg.update_moving, cfg.unbiased_moving, cfg.ws,
                           cfg.st)
        self.bn.act_fwd(self.ya, C_, self.aa, self.npix, cfg.st)
        self.conv.fwd(self.aa, self.out, cfg.st)
        return self.out

    def bwd(self, dout, cfg):
        L = self.store.L
        C_ = self.num_classes
        cfg.wgrad(self.conv, self.aa, dout)
        self.conv.bwd_data(dout, self.daa, cfg.st)
        self.bn.act_bwd(self.daa, self.ya, C_, self.dya, self.npix, cfg.ws, cfg.st)
        cfg.wgrad(self.conv_a, self.xu, self.dya)
        self.conv_a.bwd_data(self.dya, self.dxu, cfg.st)
        L.upsample_nn2x_bwd(_ptr(self.dxu), self.cin, self.n, self.h, self.w, self.cin, _ptr(self.dx), cfg.st)
        return self.dx


class ClassifierHead(Block):
    """Head of the encoder-only models (src/models.py:985-993, 1040-1048, 1347-1354, 1667-1668): average pooling over the
    whole map -> Flatten -> Dense(num_classes) named 'logits'.  The Dense layer runs as a 1x1 convolution over n pixels."""

    def __init__(self, store, cin, num_classes):
        self.name = "logits"
        self.store = store
        self.cin, self.num_classes = cin, num_classes
        self.staged = store.dtype_code != F32      # bfloat16 storage: the pooling kernels take float32, widen around them
        self.dense = Conv(store, "logits", 1, 1, cin, num_classes, padding="valid", bias=True, init="glorot_uniform",
                          kernel_shape=(cin, num_classes))

    def convs(self):
        return [self.dense]

    def plan(self, n, h, w, dev):
        f32 = dict(dtype=self.store.act_dtype, device=dev)
        self.n, self.hw = n, h * w
        self.dense.plan(n, 1, 1)
        self.pooled = torch.empty(n, 1, 1, self.cin, **f32)
        self.dpooled = torch.empty_like(self.pooled)
        self.out = torch.empty(n, self.num_classes, **f32)
        self.dout = torch.empty_like(self.out)
        self.dx = torch.empty(n, h, w, self.cin, **f32)
        if self.staged:
            w32 = dict(dtype=torch.float32, device=dev)
            self.x32 = torch.empty(n, h, w, self.cin, **w32)      # also receives the input gradient before it is rounded
            self.pooled32 = torch.empty(n, 1, 1, self.cin, **w32)
            self.dpooled32 = torch.empty_like(self.pooled32)
        return (1, 1, self.num_classes)

    def fwd(self, x, cfg):
        L = lib()
        if self.staged:
            L.cast_bf16_to_f32(_ptr(x), _ptr(self.x32), x.numel(), cfg.st)
            L.global_avgpool_fwd(_ptr(self.x32), self.n, self.hw, self.cin, _ptr(self.pooled32), cfg.st)
            L.cast_f32_to_bf16(_ptr(self.pooled32), _ptr(self.pooled), self.pooled.numel(), cfg.st)
        else:
            L.global_avgpool_fwd(_ptr(x), self.n, self.hw, self.cin, _ptr(self.pooled), cfg.st)
        self.dense.fwd(self.pooled, self.out, cfg.st)
        return self.out

    def bwd(self, dout, cfg):
        L = lib()
        cfg.wgrad(self.dense, self.pooled, dout)
        self.dense.bwd_data(dout, self.dpooled, cfg.st)
        if self.staged:
            L.cast_bf16_to_f32(_ptr(self.dpooled), _ptr(self.dpooled32), self.dpooled.numel(), cfg.st)
            L.global_avgpool_bwd(_ptr(self.dpooled32), self.n, self.hw, self.cin, _ptr(self.x32), cfg.st)
            L.cast_f32_to_bf16(_ptr(self.x32), _ptr(self.dx), self.dx.numel(), cfg.st)
        else:
            L.global_avgpool_bwd(_ptr(self.dpooled), self.n, self.hw, self.cin, _ptr(self.dx), cfg.st)
        return self.dx


class LogitsHead(Block):
    """Conv2DTranspose(num_classes, 2x2, strides 2) named 'logits' (src/models.py:1311-1317)."""

    def __init__(self, store, cin, num_classes):
        self.name = "logits"
        self.store = store
        self.conv = Conv(store, "logits", 2, 2, cin, num_classes, stride=2, bias=True, transpose=True)
        self.num_classes = num_classes

    def convs(self):
        return [self.conv]

    def plan(self, n, h, w, dev):
        self.ho, self.wo = self.conv.plan(n, h, w)
        self.out = torch.empty(n, self.ho, self.wo, self.num_classes, dtype=self.store.act_dtype, device=dev)
        self.dout = torch.empty_like(self.out)
        self.dx = torch.empty(n, h, w, self.conv.cin, dtype=self.store.act_dtype, device=dev)
        return (self.ho, self.wo, self.num_classes)

    def fwd(self, x, cfg):
        self.x = x
        self.conv.fwd(x, self.out, cfg.st)
        return self.out

    @property
    def fusable(self) -> bool:
        """The transposed convolution can run inside the loss kernel (ssmi_logits_tconv2x2_xent_fwd_bwd)."""
        return bool(lib().logits_tconv2x2_xent_supported(self.conv.cin, self.num_classes)) and \
            not os.environ.get("SSMI_NO_FUSED_HEAD")

    def xent_fused(self, x, labels, weights, npix, loss_out, cm, npix_metrics, ignore_bits, bad, ws, st, want_grad, keep_logits):
        """logits = tconv(x) and the pixelwise cross-entropy on them in ONE launch; the logits tensor is written only when
        `keep_logits`, the gradient only when `want_grad`."""
        self.x = x
        n, h, w, cin = x.shape
        lib().logits_tconv2x2_xent_fwd_bwd(self.store.dtype_code, _ptr(x), n, h, w, cin, cin, _ptr(self.conv.w), _ptr(self.conv.b),
                                           self.num_classes, _ptr(labels), _ptr(weights), npix, 1.0,
                                           _ptr(self.out) if keep_logits else None, _ptr(self.dout) if want_grad else None,
                                           _ptr(loss_out), _ptr(cm), npix_metrics, ignore_bits, None, _ptr(bad), _ptr(ws),
                                           ws.numel() * 4, st)

    def bwd(self, dout, cfg):
        cfg.wgrad(self.conv, self.x, dout)
        self.conv.bwd_data(dout, self.dx, cfg.st)
        return self.dx


# ----------------------------------------------------------------------------------------------------------
# the network
# ----------------------------------------------------------------------------------------------------------

def plan_dropout_arena(layers: List[Tuple[str, float, int]], n: int, dev):
    """SpatialDropout2D tables [n, channels] of every layer in ONE arena, grouped by rate so that one launch per
    distinct rate refreshes all of them.  Returns (arena, [(rate, offset, count)], {layer name: [n, ch] view})."""
    order = sorted(range(len(layers)), key=lambda i: layers[i][1])
    total = sum(n * layers[i][2] for i in order)
    arena = torch.ones(max(total, 4), dtype=torch.float32, device=dev)
    groups: List[Tuple[float, int, int]] = []
    views: Dict[str, torch.Tensor] = {}
    off = 0
    for i in order:
        name, rate, ch = layers[i]
        cnt = n * ch
        views[name] = arena[off:off + cnt].view(n, ch)
        if groups and groups[-1][0] == rate:
            r, o, k = groups[-1]
            groups[-1] = (r, o, k + cnt)
        else:
            groups.append((rate, off, cnt))
        off += cnt
    return arena, groups, views


ENC_PATTERN = [("1", {}), ("2", {"dilated": 2}), ("3", {"asymmetric": 5}), ("4", {"dilated": 4}),
               ("5", {}), ("6", {"dilated": 8}), ("7", {"asymmetric": 5}), ("8", {"dilated": 16})]


class GraphBase:
    """A list of blocks executed in order (forward) and in reverse (backward), the parameter arenas, the dropout-table
    arena and the reduction workspaces.  Subclasses build `self.blocks` and implement `_plan_blocks`."""

    multiple = 16   # input height/width granularity

    def _finish_init(self, blocks):
        self.blocks = blocks
        self.by_name = {b.name: b for b in blocks}
        self.store.materialise(self.device)
        self.shape = None
        self.side_stream = None
        # Weight gradients never feed the rest of the backward pass: they are collected per block and launched on a second
        # stream, so that under graph capture they form a parallel branch that fills the machine while the latency-bound
        # main chain runs (ENet 512x512, batch 16, same box: 11.79 -> 11.46 ms/step).  Must be set before the first plan().
        self.overlap_weight_gradients = True
        self.drop_layers = [d for b in blocks for d in b.dropouts()]

    # -- rotated kernel matrices for the data gradients ---------------------------------------------------
    def _plan_rotated_weights(self, dev):
        """Layers whose data gradient runs as a forward convolution over dy read a rotated, transposed copy of their
        kernel; all copies live in one arena shaped like theta and are rewritten by ONE launch per training step."""
        import numpy as _np
        st = self.store
        rows = []
        for b in self.blocks:
            for cv in b.convs():
                cv.wrot = None
                if st.act_dtype == torch.bfloat16 and (cv.pack_usage & 2):
                    continue        # its data gradient reads the bfloat16 pack instead (the float32 forms remain the fallback)
                if cv.uses_rot and not cv.transpose and dev.type == "cuda" and not os.environ.get("SSMI_NO_ROTATED_ARENA"):
                    sp = st.specs[cv.name + "/kernel"]
                    rows.append((cv, sp))
        self.rot_table = None
        self.rot_count = len(rows)
        if not rows:
            return
        if getattr(st, "theta_rot", None) is None or st.theta_rot.numel() != st.theta.numel():
            st.theta_rot = torch.zeros_like(st.theta)
        tab = _np.zeros(len(rows), dtype=_np.dtype([("offset", "<i8"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4"), ("cout", "<i4")]))
        for i, (cv, sp) in enumerate(rows):
            tab[i] = (sp.offset, cv.kh, cv.kw, cv.cin, cv.cout)
            cv.wrot = st.theta_rot[sp.offset:sp.offset + sp.size]
        self.rot_table = torch.from_numpy(tab.view(_np.uint8).copy()).to(dev)

    def _plan_weight_packs(self, dev):
        """bf16 storage: layers whose forward pass / data gradient can run on the bfloat16 matrix instructions read bfloat16
        copies of their kernels (forward form [tap][cout][cin], data-gradient form [tap'][cin][cout]); both forms of all such
        layers live in two bfloat16 arenas indexed like theta and are rewritten by ONE launch in front of every forward pass.
        The descriptors carry the addresses, so the layers need no further arguments."""
        import numpy as _np
        st = self.store
        self.pack_table = None
        self.pack_count = 0
        self.packed_layers = {}
        if st.act_dtype != torch.bfloat16 or dev.type != "cuda":
            return
        rows = []
        for b in self.blocks:
            for cv in b.convs():
                if cv.pack_usage and not cv.transpose and cv.desc is not None:
                    rows.append((cv, st.specs[cv.name + "/kernel"]))
        if not rows:
            return
        if getattr(st, "theta_fpack", None) is None or st.theta_fpack.numel() != 2 * st.theta.numel():
            st.theta_fpack = torch.zeros(2 * st.theta.numel(), dtype=torch.bfloat16, device=dev)
            st.theta_rpack = torch.zeros(2 * st.theta.numel(), dtype=torch.bfloat16, device=dev)
        tab = _np.zeros(len(rows), dtype=_np.dtype([("offset", "<i8"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4"), ("cout", "<i4")]))
        for i, (cv, sp) in enumerate(rows):
            tab[i] = (sp.offset, cv.kh, cv.kw, cv.cin, cv.cout)
            if cv.pack_usage & 1:       # a layer's pack sits at the byte offset its kernel has in theta (16-byte aligned)
                cv.desc.w_fwd_pack = st.theta_fpack.data_ptr() + 4 * sp.offset
            if cv.pack_usage & 2:
                cv.desc.w_dgrad_pack = st.theta_rpack.data_ptr() + 4 * sp.offset
            self.packed_layers[cv.name] = cv.pack_usage
        self.pack_table = torch.from_numpy(tab.view(_np.uint8).copy()).to(dev)
        self.pack_count = len(rows)

    def pack_weights(self, stream=None):
        """Refreshes the bfloat16 kernel packs from theta (bf16 storage only; one launch)."""
        if getattr(self, "pack_table", None) is None:
            return
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        lib().conv_pack_weights_bf16(_ptr(self.pack_table), self.pack_count, _ptr(self.store.theta), _ptr(self.store.theta_fpack),
                                     _ptr(self.store.theta_rpack), st)

    def rotate_weights(self, stream=None):
        """Refreshes the rotated arena from theta; to be called once per training step before the backward pass."""
        if self.rot_table is None:
            return
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        lib().conv_rotate_weights(_ptr(self.rot_table), self.rot_count, _ptr(self.store.theta), _ptr(self.store.theta_rot), st)

    # -- planning -------------------------------------------------------------------------------------
    def plan(self, n, h, w):
        if self.shape == (n, h, w):
            return
        m = self.multiple
        if h % m or w % m:
            raise ValueError("input height/width must be multiples of {} (got {}x{})".format(m, h, w))
        dev = self.device
        L = lib()
        self._plan_blocks(n, h, w, dev)
        self._plan_rotated_weights(dev)
        self._plan_weight_packs(dev)
        self.drop_arena, self.drop_groups, self.drop_views = plan_dropout_arena(self.drop_layers, n, dev)
        for b in self.blocks:
            b.bind_dropout(self.drop_views)
        # workspaces: `ws` for the main stream (BN / merge / loss partial sums, weight-gradient partials), `ws2` for
        # the weight-gradient stream
        need2 = 16
        for b in self.blocks:
            for cv in b.convs():
                need2 = max(need2, cv.ws_bytes)
        npix = n * h * w
        need = max(L.bn_workspace_bytes(0, 1024), L.softmax_xent_workspace_bytes(npix), need2)
        for b in self.blocks:
            if getattr(b, "expand", None) is not None:
                need = max(need, L.expand_workspace_bytes(C.byref(b.expand)))
        self.ws = torch.empty((need + 3) // 4 + 64, dtype=torch.float32, device=dev)
        self.ws2 = torch.empty((need2 + 3) // 4 + 64, dtype=torch.float32, device=dev) if self.overlap_weight_gradients \
            else self.ws
        # SSMI_GROUPED_REDUCE=join|block: one strip reduction per join (or per block) instead of one per layer
        # (ssmi_wg_reduce_defer_*): every layer's partial sums must then survive until that launch, so each layer gets its own
        # workspace out of one arena - where that arena stays modest (the ENets: 0.7 GB; the wide models' layers reduce long
        # rows, whose launch is not what they cost).  Off by default: bit-identical, 89 -> 2 (29) launches, and SLOWER - ENet
        # 512x512 batch 16: 7.94 (7.92) against 7.87 ms float32, 5.72 (5.67) against 5.67 bfloat16: the per-layer launches hide
        # beside the main chain, the grouped one is a tail behind the last weight gradient.
        self.defer_reduce = False
        convs = [cv for b in self.blocks for cv in b.convs()]
        for cv in convs:
            cv._ws_view = None
        total = sum((cv.ws_bytes + 255) // 256 * 256 + 256 for cv in convs)
        if (self.overlap_weight_gradients and dev.type == "cuda" and total <= (2 << 30)
                and os.environ.get("SSMI_GROUPED_REDUCE", "") in ("join", "block") and int(os.environ.get("SSMI_WGRAD_STREAMS", "1")) == 1):
            self.ws2_arena = torch.empty(total // 4, dtype=torch.float32, device=dev)
            off = 0
            for cv in convs:
                nb_ = (cv.ws_bytes + 255) // 256 * 256 + 256
                cv._ws_view = self.ws2_arena[off // 4:(off + nb_) // 4]
                off += nb_
            self.defer_reduce = True
        if self.side_stream is None and dev.type == "cuda" and self.overlap_weight_gradients:
            self.side_stream = torch.cuda.Stream(device=dev)
        # ONE weight-gradient stream.  Under graph replay the HIP runtime does not keep captured streams apart: it walks the
        # graph depth-first, keeps a node's first dependent on the node's hardware queue and gives every further dependent
        # the next queue.  One side chain whose launches are created after the main chain's next kernel (RunCfg.flush)
        # leaves the main chain on one queue and all weight gradients on the second.  Measured (ENet 512x512 batch 16,
        # float32 / bf16 storage, ms per step, one box): 1 stream 8.89 / 6.65, 2: 9.94 / 7.46, 4: 10.5 / 7.9; with the
        # launches created at their fork point (side branch = first dependent, the main chain hops from queue to queue and
        # lands behind weight gradients of earlier blocks): 4 streams 9.25 / 7.03, 1 stream 10.2 / 7.6.
        # SSMI_WGRAD_STREAMS overrides for measurements.
        nside = int(os.environ.get("SSMI_WGRAD_STREAMS", "1"))
        if self.side_stream is not None:
            check_queue_budget(nside)
        self.side_streams = self.side_wss = None
        if nside > 1 and self.side_stream is not None:
            self.side_streams = [self.side_stream] + [torch.cuda.Stream(device=dev) for _ in range(nside - 1)]
            self.side_wss = [self.ws2] + [torch.empty_like(self.ws2) for _ in range(nside - 1)]
        self.shape = (n, h, w)

    # -- execution ------------------------------------------------------------------------------------
    def forward(self, images: torch.Tensor, training: bool, update_moving: bool = True, unbiased_moving: bool = False,
                use_dropout: Optional[bool] = None, stream: Optional[int] = None, fuse_head: bool = False):
        """`fuse_head`: stop in front of a last layer that the loss kernel computes itself (LogitsHead.xent_fused); the
        return value is then that layer's input and `logits` stays unwritten unless the loss call is told to keep it."""
        n, h, w, c = images.shape
        assert c == 3 and images.dtype == torch.float32 and images.is_contiguous()
        self.plan(n, h, w)
        if self.store.dtype_code == BF16:
            # the batch arrives as float32 (src/generators.py:732-768); a bfloat16 network rounds it once, here
            if getattr(self, "_images_act", None) is None or self._images_act.shape != images.shape:
                self._images_act = torch.empty(images.shape, dtype=torch.bfloat16, device=images.device)
            stq = stream if stream is not None else torch.cuda.current_stream().cuda_stream
            lib().cast_f32_to_bf16(_ptr(images), _ptr(self._images_act), images.numel(), stq)
            images = self._images_act
        main = torch.cuda.current_stream()
        st = stream if stream is not None else main.cuda_stream
        self.pack_weights(st)       # bf16 storage: bfloat16 kernel packs of the weights this pass runs with
        side = self.side_stream if (self.overlap_weight_gradients and training and st == main.cuda_stream and st != 0) else None
        multi = side is not None and getattr(self, "side_streams", None)
        cfg = RunCfg(training, update_moving and training, unbiased_moving,
                     training if use_dropout is None else use_dropout, st, self.ws, main,
                     self.side_streams if multi else side, self.side_wss if multi else self.ws2)
        cfg.defer_reduce = bool(getattr(self, "defer_reduce", False) and side is not None and not multi)
        sto = self.store
        sto.infer_stats_batched = True
        if not training and sto.n_state:
            lib().bn_infer_stats_arena(_ptr(sto.state), _ptr(sto.var_mask), sto.n_state, BN_EPS, _ptr(sto.bn_stats), st)
        x = images
        head = self.blocks[-1]
        fused = fuse_head and isinstance(head, LogitsHead) and head.fusable
        for b in (self.blocks[:-1] if fused else self.blocks):
            x = b.fwd(x, cfg)
        self.head_input = x if fused else None
        self.logits_valid = not fused
        self._cfg = cfg
        return x

    def backward(self, dlogits: torch.Tensor):
        cfg = self._cfg
        self.rotate_weights(cfg.st)
        run_backward(reversed(self.blocks), dlogits, cfg)
        cfg.join()

    def refresh_dropout(self, seed: int, step_block: torch.Tensor, stream: Optional[int] = None):
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        L = lib()
        for gi, (rate, off, cnt) in enumerate(self.drop_groups):
            L.dropout_mask(self.drop_arena.data_ptr() + 4 * off, cnt, rate, seed, _ptr(step_block), gi, st)

    @property
    def logits(self):
        if not getattr(self, "logits_valid", True):
            raise RuntimeError("the last step computed the logits inside the loss kernel and did not store them; "
                               "set model.keep_logits = True to have them written")
        return self.blocks[-1].out

    @property
    def dlogits(self):
        return self.blocks[-1].dout


class ENetGraph(GraphBase):
    """Blocks + arenas + activation plan of `enet-naive-upsampling` / `enet-max-unpooling` /
    `enet-naive-upsampling-enhanced`."""

    multiple = 8    # three 2x reductions (initial block, bottlenecks 1.0 and 2.0)

    def __init__(self, variant: str, num_classes: int, device="cuda", encoder_only: bool = False, act_dtype=torch.float32):
        assert variant in ("naive", "unpool", "enhanced")
        self.variant, self.num_classes, self.encoder_only = variant, num_classes, encoder_only
        self.device = torch.device(device)
        st = self.store = ParamStore(act_dtype)
        B = Bottleneck
        blocks: List[Block] = [InitialBlock(st, variant)]
        blocks.append(B(st, "en_bn_1.0", 16, 64, "enc", variant, downsample=True, dropout_rate=0.01))
        for i in range(4):
            blocks.append(B(st, "en_bn_1.{}".format(i + 1), 64, 64, "enc", variant, dropout_rate=0.01))
        blocks.append(B(st, "en_bn_2.0", 64, 128, "enc", variant, downsample=True, dropout_rate=0.1))
        for stage in (2, 3):
            for suffix, kw in ENC_PATTERN:
                blocks.append(B(st, "en_bn_{}.{}".format(stage, suffix), 128, 128, "enc", variant, dropout_rate=0.1, **kw))
        if encoder_only and variant != "enhanced":      # src/models.py:1040-1048, 1346-1354: the head sits on the encoder
            blocks.append(ClassifierHead(st, 128, num_classes))
            self._finish_init(blocks)
            return
        ddrop = None if variant == "unpool" else 0.1   # the max-unpooling decoder has no dropout (src/models.py:1576-1607)
        blocks.append(B(st, "de_bn_4.0", 128, 64, "dec", variant, upsample=True, dropout_rate=ddrop))
        blocks.append(B(st, "de_bn_4.1", 64, 64, "dec", variant, dropout_rate=ddrop))
        blocks.append(B(st, "de_bn_4.2", 64, 64, "dec", variant, dropout_rate=ddrop))
        blocks.append(B(st, "de_bn_5.0", 64, 16, "dec", variant, upsample=True, dropout_rate=ddrop))
        blocks.append(B(st, "de_bn_5.1", 16, 16, "dec", variant, dropout_rate=ddrop))
        if encoder_only:      # enhanced: the WHOLE network, its last layer renamed, then the classifier (src/models.py:1665-1668)
            blocks.append(EnhancedHead(st, 16, num_classes, final_layer_name="decoder_logits"))
            blocks.append(ClassifierHead(st, num_classes, num_classes))
        else:
            blocks.append(EnhancedHead(st, 16, num_classes) if variant == "enhanced" else LogitsHead(st, 16, num_classes))
        self._finish_init(blocks)

    def _plan_blocks(self, n, h, w, dev):
        hh, ww, c = self.blocks[0].plan(n, h, w, dev)
        for b in self.blocks[1:]:
            if isinstance(b, Bottleneck):
                uidx = None
                if b.mode == 3:
                    src = self.by_name["en_bn_2.0" if b.name == "de_bn_4.0" else "en_bn_1.0"]
                    uidx = src.idx   # LIFO index stack of src/models.py:1614,1617
                hh, ww, c = b.plan(n, hh, ww, dev, unpool_idx=uidx)
            else:
                hh, ww, c = b.plan(n, hh, ww, dev)
        # A bottleneck that recomputes its expansion carries its merge forward through the 1x1 reduction the next regular
        # bottleneck starts with (ssmi_expand_merge_fwd_through).  SSMI_NO_FWD_THROUGH=1: the reduction as its own launch.
        for a, b in zip(self.blocks, self.blocks[1:]):
            if isinstance(b, Bottleneck):
                b.prev_block = a if isinstance(a, Bottleneck) else None
            if isinstance(a, Bottleneck):
                a.next_fwd = None
                if (isinstance(b, Bottleneck) and a.expand is not None and b.mode == 0 and b.cin == a.cout and b.internal == a.expand.cin
                        and (b.hin, b.win) == (a.ho, a.wo) and not os.environ.get("SSMI_NO_FWD_THROUGH")):
                    a.next_fwd = b
