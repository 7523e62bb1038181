"""MaxPoolingWithArgmax2D / MaxUnpooling2D with the constructor and call shapes of the reference's
src/layers/pooling.py, running on libssmi.so.  Inputs and outputs are NHWC CUDA (ROCm) torch tensors; both layers are
differentiable through torch.autograd (their backward passes are HIP kernels too)."""
from __future__ import annotations

import torch

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(x, who):
    if not x.is_cuda:
        # the reference raises the same error type when TensorFlow sees no GPU (src/layers/pooling.py:26-27)
        raise NotImplementedError('{} works only on GPU'.format(who))


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, h, w, c = x.shape
        ho, wo = (h + 1) // 2, (w + 1) // 2
        y = torch.empty(n, ho, wo, c, dtype=torch.float32, device=x.device)
        idx = torch.empty(n, ho, wo, c, dtype=torch.int32, device=x.device)
        idx_f = torch.empty(n, ho, wo, c, dtype=torch.float32, device=x.device)
        lib().maxpool_argmax_fwd(x.data_ptr(), n, h, w, c, y.data_ptr(), c, idx.data_ptr(), idx_f.data_ptr(), _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, h, w, c)
        ctx.mark_non_differentiable(idx_f)
        return y, idx_f

    @staticmethod
    def backward(ctx, dy, _didx):
        (idx,) = ctx.saved_tensors
        n, h, w, c = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
        lib().maxpool_argmax_bwd(dy.data_ptr(), c, idx.data_ptr(), n, h, w, c, dx.data_ptr(), 0, _stream())
        return dx


class _UnpoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx_f):
        n, h, w, c = x.shape
        idx = torch.empty(n, h, w, c, dtype=torch.int32, device=x.device)
        L = lib()
        L.idx_f32_to_i32(idx_f.data_ptr(), idx.data_ptr(), idx.numel(), _stream())
        y = torch.empty(n, 2 * h, 2 * w, c, dtype=torch.float32, device=x.device)
        L.maxunpool_fwd(x.data_ptr(), idx.data_ptr(), n, h, w, c, y.data_ptr(), _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, h, w, c = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
        lib().maxunpool_bwd(dy.data_ptr(), idx.data_ptr(), n, h, w, c, dx.data_ptr(), _stream())
        return dx, None


class MaxPoolingWithArgmax2D(object):
    """src/layers/pooling.py:7-58.  Returns [pooled, argmax] with argmax as float32 per-image flat NHWC indices."""

    def __init__(self, pool_size=(2, 2), strides=(2, 2), padding='same', name=None, **kwargs):
        if tuple(pool_size) != (2, 2) or tuple(strides) != (2, 2) or padding.lower() != 'same':
            raise NotImplementedError('only pool_size=(2,2), strides=(2,2), padding="same" is supported')
        self.pool_size, self.strides, self.padding, self.name = pool_size, strides, padding, name

    def __call__(self, inputs, **kwargs):
        _require_gpu(inputs, type(self).__name__)
        y, idx_f = _PoolFn.apply(inputs.contiguous().float())
        return [y, idx_f]

    def compute_output_shape(self, input_shape):
        out = tuple(None if d is None else (d if i in (0, 3) else (d + 1) // 2) for i, d in enumerate(input_shape))
        return [out, out]


class MaxUnpooling2D(object):
    """src/layers/pooling.py:61-112.  call([values, argmax]) scatters the values into a zero tensor of twice the size."""

    def __init__(self, size=(2, 2), name=None, **kwargs):
        if tuple(size) != (2, 2):
            raise NotImplementedError('only size=(2,2) is supported')
        self.size, self.name = size, name

    def __call__(self, inputs, output_shape=None):
        pool, ind = inputs[0], inputs[1]
        _require_gpu(pool, type(self).__name__)
        if tuple(pool.shape) != tuple(ind.shape):
            raise ValueError('values {} and indices {} must have the same shape'.format(tuple(pool.shape), tuple(ind.shape)))
        return _UnpoolFn.apply(pool.contiguous().float(), ind.contiguous().float())

    def compute_output_shape(self, input_shape):
        m = input_shape[1]
        return (m[0], None if m[1] is None else m[1] * self.size[0], None if m[2] is None else m[2] * self.size[1], m[3])
