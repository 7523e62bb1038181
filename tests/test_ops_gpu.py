"""Per-operator GPU parity through the C ABI against the CPU oracle (fp64), rtol 1e-3 for floating point,
bit-exact for indices and confusion matrices.  Shapes cover the kernel configurations of the
ENet graphs plus ragged / odd / tie cases."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_keras_ops as K

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def _check(got, ref, what, rtol=1e-3, atol_rel=1e-4):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().numpy().astype(np.float64) if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, "{}: shape {} vs {}".format(what, got.shape, ref.shape)
    scale = np.abs(ref).max() if ref.size else 0.0
    err = np.abs(got - ref)
    bad = err > atol_rel * scale + rtol * np.abs(ref) + 1e-12
    assert not bad.any(), "{}: {} of {} off, max err {:.3e}, scale {:.3e}".format(what, int(bad.sum()), ref.size, float(err.max()), scale)


CONV_CASES = [
    # kh kw cin cout stride dil padding bias  N  H  W
    (3, 3, 3, 13, 2, 1, "same", True, 2, 16, 20),      # initial block conv (scalar path, cout < 32)
    (2, 2, 16, 16, 2, 1, "valid", False, 2, 12, 8),    # downsampling projection
    (2, 2, 64, 32, 2, 1, "valid", False, 1, 8, 8),
    (1, 1, 64, 16, 1, 1, "valid", False, 2, 9, 7),     # ragged pixel tiles
    (1, 1, 16, 64, 1, 1, "valid", False, 1, 8, 8),
    (1, 1, 32, 128, 1, 1, "valid", False, 1, 8, 8),    # 4 column tiles
    (1, 1, 128, 32, 1, 1, "valid", False, 1, 6, 6),
    (3, 3, 16, 16, 1, 1, "same", True, 2, 8, 10),
    (3, 3, 32, 32, 1, 1, "same", True, 1, 9, 9),
    (3, 3, 32, 32, 1, 2, "same", True, 1, 8, 8),
    (3, 3, 32, 32, 1, 16, "same", True, 1, 8, 8),      # dilation larger than the map
    (1, 5, 32, 32, 1, 1, "same", False, 1, 6, 11),
    (5, 1, 32, 32, 1, 1, "same", True, 1, 11, 6),
    (3, 3, 4, 4, 1, 1, "same", True, 2, 8, 8),
    (1, 1, 4, 16, 1, 1, "valid", False, 1, 8, 8),
    (1, 1, 64, 4, 1, 1, "valid", False, 1, 8, 8),
    (1, 1, 16, 4, 1, 1, "valid", False, 2, 9, 7),      # vector-arithmetic kernel, four operand loads per pixel
    (3, 3, 4, 4, 1, 1, "same", True, 2, 384, 352),     # the same kernel past its grid cap: several pixels per thread
    (3, 3, 5, 7, 2, 1, "same", True, 1, 7, 9),         # odd everything
    (3, 3, 24, 24, 1, 1, "same", True, 1, 8, 8),       # enhanced head width
    (3, 3, 64, 64, 1, 1, "same", True, 2, 9, 11),      # wide layers: tiled weight gradient (256 x 64 tiles), ragged pixels
    (3, 3, 32, 128, 1, 1, "same", False, 1, 8, 8),     # 128 x 128 tiles, K = 288 (three k-tiles, the last one partial)
    (1, 1, 256, 64, 1, 1, "valid", True, 1, 7, 9),
    (3, 3, 48, 192, 1, 2, "same", True, 1, 10, 6),     # cout = 3 x 64, dilated taps
    (3, 3, 136, 128, 1, 1, "same", True, 3, 5, 7),     # k-tile spanning two taps (cin not a multiple of the tile)
    # 1x1 layers of the direct weight-gradient kernel (operands in the matrix instruction's layout): every channel pair class,
    # pixel counts that are no multiple of the four-pixel groups, several workgroups and several rounds per wave
    (1, 1, 128, 64, 1, 1, "valid", False, 2, 37, 29),
    (1, 1, 128, 16, 1, 1, "valid", False, 3, 21, 23),
    (1, 1, 64, 64, 1, 1, "valid", False, 1, 50, 41),
    (1, 1, 32, 32, 1, 1, "valid", False, 2, 19, 17),
    (1, 1, 16, 16, 1, 1, "valid", False, 1, 45, 47),
    (1, 1, 64, 128, 1, 1, "valid", False, 1, 33, 31),
    (1, 1, 32, 128, 1, 1, "valid", False, 4, 64, 64),
    (1, 1, 128, 32, 1, 1, "valid", False, 4, 64, 64),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}".format(*c[:7]))
def test_conv2d_fwd_bwd(case):
    kh, kw, cin, cout, s, d, pad, bias, N, H, W = case
    eng = pkg("engine")
    rng = np.random.RandomState(hash(case) % 2**31)
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=s, dil=d, padding=pad, bias=bias)
    store.materialise(torch.device(DEV))
    w = rng.standard_normal((kh, kw, cin, cout))
    b = rng.standard_normal(cout) if bias else None
    store.views["c/kernel"].copy_(_t(w))
    if bias:
        store.views["c/bias"].copy_(_t(b))
    ho, wo = cv.plan(N, H, W)
    x = rng.standard_normal((N, H, W, cin))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True) if bias else None
    yr = K.conv2d(xt, wt, bt, s, d, pad)
    assert tuple(yr.shape) == (N, ho, wo, cout)
    dy = rng.standard_normal(tuple(yr.shape))
    add = rng.standard_normal(x.shape)
    yr.backward(torch.tensor(dy))
    st = torch.cuda.current_stream().cuda_stream
    xd, dyd = _t(x), _t(dy)
    y = torch.empty(N, ho, wo, cout, device=DEV)
    cv.fwd(xd, y, st)
    _check(y, yr, "y")
    dx = torch.full((N, H, W, cin), 7.0, device=DEV)
    cv.bwd_data(dyd, dx, st)
    _check(dx, xt.grad, "dx")
    cv.bwd_data(dyd, dx, st, add=_t(add))
    _check(dx, xt.grad + torch.tensor(add), "dx+add")
    ws = torch.empty(cv.ws_bytes // 4 + 8, device=DEV)
    cv.bwd_weight(xd, dyd, ws, st)
    _check(store.gviews["c/kernel"], wt.grad, "dw")
    if bias:
        _check(store.gviews["c/bias"], bt.grad, "db")


@pytest.mark.parametrize("k,cin,cout,N,H,W", [(3, 16, 16, 2, 6, 5), (3, 4, 4, 1, 8, 8), (2, 16, 24, 2, 5, 7), (2, 16, 5, 1, 4, 4), (2, 16, 24, 1, 3, 40), (2, 32, 64, 2, 4, 33)])
def test_conv2d_transpose_fwd_bwd(k, cin, cout, N, H, W):
    eng = pkg("engine")
    rng = np.random.RandomState(k * 100 + cin)
    store = eng.ParamStore()
    cv = eng.Conv(store, "t", k, k, cin, cout, stride=2, bias=True, transpose=True)
    store.materialise(torch.device(DEV))
    w = rng.standard_normal((k, k, cout, cin))
    b = rng.standard_normal(cout)
    store.views["t/kernel"].copy_(_t(w))
    store.views["t/bias"].copy_(_t(b))
    ho, wo = cv.plan(N, H, W)
    assert (ho, wo) == (2 * H, 2 * W)
    x = rng.standard_normal((N, H, W, cin))
    xt = torch.tensor(x, requires_grad=True)
    wt = torch.tensor(w, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    yr = K.conv2d_transpose(xt, wt, bt, 2)
    dy = rng.standard_normal(tuple(yr.shape))
    yr.backward(torch.tensor(dy))
    st = torch.cuda.current_stream().cuda_stream
    y = torch.empty(N, ho, wo, cout, device=DEV)
    cv.fwd(_t(x), y, st)
    _check(y, yr, "yt")
    dx = torch.empty(N, H, W, cin, device=DEV)
    cv.bwd_data(_t(dy), dx, st)
    _check(dx, xt.grad, "dxt")
    ws = torch.empty(cv.ws_bytes // 4 + 8, device=DEV)
    cv.bwd_weight(_t(x), _t(dy), ws, st)
    _check(store.gviews["t/kernel"], wt.grad, "dw")
    _check(store.gviews["t/bias"], bt.grad, "db")


@pytest.mark.parametrize("c,npix,act", [(16, 1000, 1), (4, 333, 1), (128, 260, 1), (64, 77, 0), (24, 50, 2)])
def test_bn_act_fwd_bwd(c, npix, act):
    L = pkg("_lib").lib()
    rng = np.random.RandomState(c + npix)
    x = rng.standard_normal((1, 1, npix, c)) * 2 + 0.5
    gamma, beta = rng.uniform(0.5, 1.5, c), rng.uniform(-0.5, 0.5, c)
    alpha = rng.uniform(0.05, 0.4, c)
    mm, mv = rng.standard_normal(c), rng.uniform(0.5, 2, c)
    xt = torch.tensor(x, requires_grad=True)
    gt, bt, at = (torch.tensor(v, requires_grad=True) for v in (gamma, beta, alpha))
    z, mean, var = K.batch_norm_train(xt, gt, bt)
    yr = K.prelu(z, at) if act == 1 else (K.leaky_relu(z, 0.1) if act == 2 else z)
    dy = rng.standard_normal(x.shape)
    yr.backward(torch.tensor(dy))
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(L.bn_workspace_bytes(npix, c) // 4 + 8, device=DEV)
    xd = _t(x)
    mmd, mvd = _t(mm), _t(mv)
    mean_d, inv_d = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    L.bn_train_stats(xd.data_ptr(), npix, c, c, 1e-3, 0.1, 0, mmd.data_ptr(), mvd.data_ptr(), mean_d.data_ptr(), inv_d.data_ptr(),
                     ws.data_ptr(), ws.numel() * 4, st)
    _check(mean_d, mean, "mean")
    _check(inv_d, torch.rsqrt(var + 1e-3), "invstd")
    _check(mmd, 0.1 * mm + 0.9 * mean.detach().numpy(), "moving_mean")
    _check(mvd, 0.1 * mv + 0.9 * var.detach().numpy(), "moving_var")
    g_d, b_d, a_d = _t(gamma), _t(beta), _t(alpha)
    y = torch.empty(npix, c, device=DEV)
    L.bn_act_fwd(xd.data_ptr(), c, mean_d.data_ptr(), inv_d.data_ptr(), g_d.data_ptr(), b_d.data_ptr(), act, a_d.data_ptr(), 0.1,
                 y.data_ptr(), npix, c, st)
    _check(y.view(1, 1, npix, c), yr, "y")
    dx, dg, db, da = (torch.empty(s, device=DEV) for s in ((npix, c), (c,), (c,), (c,)))
    L.bn_act_bwd(_t(dy).data_ptr(), xd.data_ptr(), c, mean_d.data_ptr(), inv_d.data_ptr(), g_d.data_ptr(), b_d.data_ptr(), act,
                 a_d.data_ptr(), 0.1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), da.data_ptr(), npix, c, ws.data_ptr(),
                 ws.numel() * 4, st)
    _check(dx.view(1, 1, npix, c), xt.grad, "dx")
    _check(dg, gt.grad, "dgamma")
    _check(db, bt.grad, "dbeta")
    if act == 1:
        _check(da, at.grad, "dalpha")
    # unbiased moving variance flag + inference statistics
    mvd2 = _t(mv)
    L.bn_train_stats(xd.data_ptr(), npix, c, c, 1e-3, 0.1, 1, None, mvd2.data_ptr(), mean_d.data_ptr(), inv_d.data_ptr(),
                     ws.data_ptr(), ws.numel() * 4, st)
    _check(mvd2, 0.1 * mv + 0.9 * var.detach().numpy() * npix / (npix - 1), "moving_var unbiased")
    L.bn_infer_stats(_t(mm).data_ptr(), _t(mv).data_ptr(), c, 1e-3, mean_d.data_ptr(), inv_d.data_ptr(), st)
    _check(inv_d, 1 / np.sqrt(mv + 1e-3), "infer invstd")


@pytest.mark.parametrize("mode,c,c_other,drop", [(0, 64, 64, True), (0, 16, 16, False), (1, 64, 16, True), (1, 128, 64, True),
                                                (2, 64, 64, True), (2, 16, 16, False), (3, 64, 64, False), (3, 16, 16, False)])
def test_merge_fwd_bwd(mode, c, c_other, drop):
    lib_mod = pkg("_lib")
    L = lib_mod.lib()
    rng = np.random.RandomState(mode * 1000 + c)
    n, h, w = 2, 6, 8
    main = rng.standard_normal((n, h, w, c))
    g3, b3 = rng.uniform(0.5, 1.5, c), rng.uniform(-0.3, 0.3, c)
    alpha = rng.uniform(0.05, 0.4, c)
    dropm = (rng.uniform(size=(n, c)) > 0.3) / 0.7 if drop else None
    mt = torch.tensor(main, requires_grad=True)
    g3t, b3t, at = (torch.tensor(v, requires_grad=True) for v in (g3, b3, alpha))
    z3, mean3, var3 = K.batch_norm_train(mt, g3t, b3t)
    if drop:
        z3 = K.spatial_dropout(z3, torch.tensor(dropm))
    idx_np = None
    if mode == 0:
        other = rng.standard_normal((n, h, w, c))
        ot = torch.tensor(other, requires_grad=True)
        o2 = ot
    elif mode == 1:
        other = rng.standard_normal((n, h, w, c_other))
        ot = torch.tensor(other, requires_grad=True)
        o2 = K.channel_zero_pad(ot, c)
    else:
        other = rng.standard_normal((n, h // 2, w // 2, c))
        go, bo = rng.uniform(0.5, 1.5, c), rng.uniform(-0.3, 0.3, c)
        ot = torch.tensor(other, requires_grad=True)
        got_, bot_ = torch.tensor(go, requires_grad=True), torch.tensor(bo, requires_grad=True)
        zo, mean_o, var_o = K.batch_norm_train(ot, got_, bot_)
        zo.retain_grad()
        if mode == 2:
            o2 = K.upsample_nn2(zo)
        else:
            src = rng.standard_normal((n, h, w, c)).astype(np.float32)
            _, idx_np = K.max_pool_with_argmax_np(src)
            o2 = K.max_unpool(zo, torch.from_numpy(idx_np))
    sr = z3 + o2
    outr = K.prelu(sr, at)
    dout = rng.standard_normal((n, h, w, c))
    outr.backward(torch.tensor(dout))

    st = torch.cuda.current_stream().cuda_stream
    md = lib_mod.MergeDesc(n, h, w, c, mode, c_other)
    bn3 = lib_mod.BnParams(*(t.data_ptr() for t in (_t(mean3.detach().numpy()), _t(torch.rsqrt(var3 + 1e-3).detach().numpy()), _t(g3), _t(b3))))
    keep = []

    def P(a, dtype=torch.float32):
        t = _t(a, dtype)
        keep.append(t)
        return t
    bn3 = lib_mod.BnParams(P(mean3.detach().numpy()).data_ptr(), P(torch.rsqrt(var3 + 1e-3).detach().numpy()).data_ptr(),
                           P(g3).data_ptr(), P(b3).data_ptr())
    bno = None
    if mode >= 2:
        bno = lib_mod.BnParams(P(mean_o.detach().numpy()).data_ptr(), P(torch.rsqrt(var_o + 1e-3).detach().numpy()).data_ptr(),
                               P(go).data_ptr(), P(bo).data_ptr())
    main_d, other_d, alpha_d = P(main), P(other), P(alpha)
    drop_d = P(dropm) if drop else None
    idx_d = P(idx_np, torch.int32) if idx_np is not None else None
    out = torch.empty(n, h, w, c, device=DEV)
    L.merge_fwd(C.byref(md), main_d.data_ptr(), C.byref(bn3), None if drop_d is None else drop_d.data_ptr(), other_d.data_ptr(),
                None if bno is None else C.byref(bno), None if idx_d is None else idx_d.data_ptr(), alpha_d.data_ptr(),
                out.data_ptr(), st)
    _check(out, outr, "out")
    ws = torch.empty(L.merge_workspace_bytes(C.byref(md)) // 4 + 8, device=DEV)
    dmain, dsum = torch.empty_like(out), torch.empty_like(out)
    dother = torch.empty(n, h // 2, w // 2, c, device=DEV) if mode >= 2 else None
    dg, db, da = (torch.empty(c, device=DEV) for _ in range(3))
    L.merge_bwd(C.byref(md), P(dout).data_ptr(), main_d.data_ptr(), C.byref(bn3), None if drop_d is None else drop_d.data_ptr(),
                other_d.data_ptr(), None if bno is None else C.byref(bno), None if idx_d is None else idx_d.data_ptr(),
                alpha_d.data_ptr(), dmain.data_ptr(), dsum.data_ptr(), None if dother is None else dother.data_ptr(),
                dg.data_ptr(), db.data_ptr(), da.data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
    _check(dmain, mt.grad, "dmain")
    _check(dg, g3t.grad, "dgamma3")
    _check(db, b3t.grad, "dbeta3")
    _check(da, at.grad, "dalpha")
    if mode == 0:
        _check(dsum, ot.grad, "dsum")
    elif mode == 1:
        _check(dsum[..., :c_other], ot.grad, "dsum[:c_other]")
    else:
        _check(dother, zo.grad, "dother_bn")


@pytest.mark.parametrize("n,h,w,c", [(2, 8, 8, 16), (1, 7, 9, 3), (2, 6, 4, 64), (1, 2, 2, 1), (1, 5, 5, 4)])
def test_maxpool_argmax_unpool_bit_exact(n, h, w, c):
    pooling = pkg("layers.pooling")
    rng = np.random.RandomState(h * 10 + c)
    x = rng.randint(-3, 4, size=(n, h, w, c)).astype(np.float32)      # small integers -> many ties
    yr, ir = K.max_pool_with_argmax_np(x)
    xd = _t(x).requires_grad_(True)
    y, idx_f = pooling.MaxPoolingWithArgmax2D()(xd)
    assert idx_f.dtype == torch.float32
    assert np.array_equal(y.detach().cpu().numpy(), yr)
    assert np.array_equal(idx_f.cpu().numpy().astype(np.int64), ir)
    dy = rng.standard_normal(yr.shape).astype(np.float32)
    y.backward(_t(dy))
    dxr = np.zeros((n, h * w * c), dtype=np.float32)
    np.add.at(dxr, (np.arange(n)[:, None], ir.reshape(n, -1)), dy.reshape(n, -1))
    assert np.array_equal(xd.grad.cpu().numpy().reshape(n, -1), dxr)
    if h % 2 == 0 and w % 2 == 0:
        vals = _t(rng.standard_normal(yr.shape).astype(np.float32)).requires_grad_(True)
        up = pooling.MaxUnpooling2D()([vals, idx_f])
        upr = K.max_unpool_np(vals.detach().cpu().numpy(), ir)
        assert np.array_equal(up.detach().cpu().numpy(), upr)
        g = rng.standard_normal(upr.shape).astype(np.float32)
        up.backward(_t(g))
        assert np.array_equal(vals.grad.cpu().numpy().reshape(n, -1), np.take_along_axis(g.reshape(n, -1), ir.reshape(n, -1), 1))
        # window-local fast path gives the same tensor
        L = pkg("_lib").lib()
        idx_i = _t(ir, torch.int32)
        out = torch.empty(n, h, w, c, device=DEV)
        L.maxunpool_local_fwd(vals.data_ptr(), idx_i.data_ptr(), n, h // 2, w // 2, c, out.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(out.cpu().numpy(), upr)


def test_maxunpool_general_indices_sum_duplicates():
    pooling = pkg("layers.pooling")
    vals = _t(np.array([[[[1.0], [2.0]], [[3.0], [4.0]]]], dtype=np.float32))        # [1,2,2,1]
    idx = _t(np.array([[[[5.0], [5.0]], [[0.0], [15.0]]]], dtype=np.float32))
    out = pooling.MaxUnpooling2D()([vals, idx]).cpu().numpy().reshape(-1)
    exp = np.zeros(16, dtype=np.float32)
    exp[5], exp[0], exp[15] = 3.0, 3.0, 4.0          # tf.scatter_nd sums duplicates
    assert np.array_equal(out, exp)


@pytest.mark.parametrize("C_,npix", [(24, 5000), (24, 77), (5, 1000), (33, 300), (2, 64)])
def test_softmax_xent_loss_grad_cm(C_, npix):
    losses = pkg("losses")
    metrics = pkg("metrics")
    rng = np.random.RandomState(C_ + npix)
    logits = rng.standard_normal((1, 1, npix, C_)) * 3
    labels = rng.randint(0, C_, size=(1, 1, npix))
    cw = rng.uniform(0.5, 2.0, C_)
    cw[0] = 0
    pw = cw[labels]
    zt = torch.tensor(logits, requires_grad=True)
    lr = K.sparse_weighted_pixelwise_xent(zt, torch.tensor(labels), torch.tensor(pw))
    lr.backward()
    loss = losses.segmentation_categorical_cross_entropy_lambda_loss(
        [_t(logits), _t(labels, torch.int32), _t(pw), torch.zeros(1, 1, dtype=torch.int32, device=DEV)])
    _check(loss, lr, "loss")
    _check(losses.last_gradient(), zt.grad, "dlogits")
    f = losses.segmentation_sparse_weighted_categorical_cross_entropy(cw)
    _check(f(_t(labels[..., None], torch.int32), _t(logits)), lr, "class-weight form")
    assert losses.dummy_loss(None, loss) is loss
    # confusion matrix: bit-exact against numpy on the float32 logits the kernel sees, cumulative, resettable
    l32 = logits.astype(np.float32)
    cmf = metrics.segmentation_confusion_matrix(C_, 0, [0])
    cm1 = cmf(_t(labels[..., None], torch.int32), _t(l32))
    ref = K.confusion_matrix_np(labels, l32, C_, [0])
    assert cm1.dtype == np.int64 and np.array_equal(cm1, ref)
    cm2 = cmf(_t(labels[..., None], torch.int32), _t(l32))
    assert np.array_equal(cm2, 2 * ref)
    cmf.reset_op()
    assert np.array_equal(cmf(_t(labels[..., None], torch.int32), _t(l32)), ref)
    acc = metrics.segmentation_accuracy(0, [0])(_t(labels[..., None], torch.int32), _t(l32))
    assert abs(acc - K.cfm_accuracy(ref)) < 1e-12
    miou = metrics.segmentation_mean_iou(C_, 0, [0])(_t(labels[..., None], torch.int32), _t(l32))
    assert abs(miou - K.tf_mean_iou(ref)) < 1e-12


def test_softmax_xent_argmax_ties_and_extreme_logits():
    metrics = pkg("metrics")
    losses = pkg("losses")
    logits = np.zeros((1, 1, 4, 24), dtype=np.float32)
    logits[0, 0, 1, [3, 7]] = 2.0            # tie -> lowest index 3
    logits[0, 0, 2, 23] = 80.0               # large but finite: stable softmax
    logits[0, 0, 3, :] = -1e4
    labels = np.array([[[5, 3, 23, 1]]], dtype=np.int32)
    cm = metrics.segmentation_confusion_matrix(24, 0, None)(_t(labels[..., None], torch.int32), _t(logits))
    assert cm[5, 0] == 1 and cm[3, 3] == 1 and cm[23, 23] == 1 and cm[1, 0] == 1 and cm.sum() == 4
    loss = losses.segmentation_categorical_cross_entropy_lambda_loss(
        [_t(logits), _t(labels, torch.int32), torch.ones(1, 1, 4, device=DEV), torch.zeros(1, 1, dtype=torch.int32, device=DEV)])
    ref = K.sparse_weighted_pixelwise_xent(torch.tensor(logits, dtype=torch.float64), torch.tensor(labels), torch.ones(1, 1, 4, dtype=torch.float64))
    _check(loss, ref, "loss")
    assert np.isfinite(losses.last_gradient().cpu().numpy()).all()


@pytest.mark.parametrize("kind", ["adam", "adam_amsgrad_decay_scalers", "sgd", "sgd_nesterov_decay"])
def test_optimizers_three_steps(kind):
    eng = pkg("engine")
    xo = pkg("keras_extensions.extended_optimizers")
    rng = np.random.RandomState(7)
    store = eng.ParamStore()
    shapes = [(3, 3, 4, 5), (5,), (1, 1, 7), (13,)]
    for i, s in enumerate(shapes):
        store.add("p{}".format(i), s)
    store.materialise(torch.device(DEV))
    p0 = [rng.standard_normal(s) for s in shapes]
    for i, v in enumerate(p0):
        store.views["p{}".format(i)].copy_(_t(v))
    scal = {1: 0.5, 3: 2.0}
    if kind == "adam":
        opt, kw = xo.Adam(lr=1e-2), dict(lr=1e-2)
    elif kind == "adam_amsgrad_decay_scalers":
        opt, kw = xo.Adam(lr=1e-2, decay=0.1, amsgrad=True, lr_scalers=scal), dict(lr=1e-2, decay=0.1, lr_scalers=scal)
    elif kind == "sgd":
        opt, kw = xo.SGD(lr=0.1, momentum=0.9), dict(lr=0.1, momentum=0.9)
    else:
        opt, kw = xo.SGD(lr=0.1, momentum=0.8, decay=0.05, nesterov=True, lr_scalers=scal), dict(lr=0.1, momentum=0.8, decay=0.05, nesterov=True, lr_scalers=scal)
    opt.bind(store)
    ps = [torch.tensor(v) for v in p0]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    vh = [torch.zeros_like(p) for p in ps] if "amsgrad" in kind else None
    for it in range(3):
        gs = [rng.standard_normal(s) for s in shapes]
        for i, g in enumerate(gs):
            store.gviews["p{}".format(i)].copy_(_t(g))
        opt.apply(grad_scale=0.5)
        gt = [torch.tensor(g) * 0.5 for g in gs]
        if kind.startswith("adam"):
            K.adam_step(ps, gt, ms, vs, it, vhats=vh, **kw)
        else:
            K.sgd_step(ps, gt, ms, it, **kw)
        assert opt.iterations == it + 1
        for i, p in enumerate(ps):
            _check(store.views["p{}".format(i)], p, "{} step {} p{}".format(kind, it, i), rtol=1e-4, atol_rel=1e-5)


def test_dropout_table_statistics_and_replay():
    L = pkg("_lib").lib()
    n = 1 << 16
    m = torch.empty(n, device=DEV)
    step = torch.zeros(2, dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.dropout_mask(m.data_ptr(), n, 0.1, 14874, step.data_ptr(), 0, st)
    a = m.cpu().numpy()
    assert set(np.unique(a)).issubset({0.0, np.float32(1 / 0.9)})
    assert abs((a == 0).mean() - 0.1) < 0.01
    L.dropout_mask(m.data_ptr(), n, 0.1, 14874, step.data_ptr(), 0, st)
    assert np.array_equal(m.cpu().numpy(), a)                 # same (seed, step, stream) -> same table
    step[0] = 1
    L.dropout_mask(m.data_ptr(), n, 0.1, 14874, step.data_ptr(), 0, st)
    assert (m.cpu().numpy() != a).mean() > 0.1                # next step -> fresh draw
    L.dropout_mask(m.data_ptr(), n, 0.0, 1, step.data_ptr(), 3, st)
    assert (m.cpu().numpy() == 1).all()


def test_error_reporting_through_the_abi():
    lib_mod = pkg("_lib")
    L = lib_mod.lib()
    with pytest.raises(lib_mod.SsmiError, match="num_classes"):
        L.softmax_xent_fwd_bwd(1, 1, 1, 10, 100, 1.0, None, 1, None, 0, 0, None, None, 1, 1 << 20, None)
    with pytest.raises(lib_mod.SsmiError, match="multiple of 4"):
        L.bn_act_fwd(1, 6, 1, 1, 1, 1, 0, None, 0.0, 1, 10, 6, None)
    d = lib_mod.ConvDesc(1, 4, 4, 3, 4, 4, 3, 3, 3, 1, 1, 1, 1, 2, 3)     # x_ld < cin
    with pytest.raises(lib_mod.SsmiError, match="leading dimension"):
        L.conv2d_fwd(C.byref(d), 1, 1, None, 1, None)


@pytest.mark.parametrize("n,h,w,c,ld", [(2, 5, 7, 16, 16), (1, 4, 4, 8, 24), (2, 3, 5, 3, 7), (1, 8, 8, 64, 96)])
def test_upsample_nn2x_into_channel_slice(n, h, w, c, ld):
    """UpSampling2D(2) forward writes a channel slice of a wider buffer (the concat of the U-Net decoder is never
    materialised); backward sums each 2x2 window.  Exact: copies and 4-term sums in a fixed order."""
    L = pkg("_lib").lib()
    rng = np.random.RandomState(n * 100 + c)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    off = ld - c
    y = torch.full((n, 2 * h, 2 * w, ld), -7.0, device=DEV)
    xd = _t(x)
    L.upsample_nn2x_fwd(xd.data_ptr(), n, h, w, c, y.data_ptr() + 4 * off, ld, st)
    ref = K.upsample_nn2(torch.from_numpy(x)).numpy()
    got = y.cpu().numpy()
    assert np.array_equal(got[..., off:], ref)
    assert (got[..., :off] == -7.0).all()                       # the neighbouring slice is untouched
    dy = rng.standard_normal((n, 2 * h, 2 * w, ld)).astype(np.float32)
    dx = torch.empty(n, h, w, c, device=DEV)
    dyd = _t(dy)
    L.upsample_nn2x_bwd(dyd.data_ptr() + 4 * off, ld, n, h, w, c, dx.data_ptr(), st)
    d = dy[..., off:].astype(np.float64)
    ref_dx = d[:, 0::2, 0::2] + d[:, 0::2, 1::2] + d[:, 1::2, 0::2] + d[:, 1::2, 1::2]
    _check(dx, ref_dx, "upsample bwd", rtol=1e-6, atol_rel=1e-6)


@pytest.mark.parametrize("npix,c,sld,dld", [(37, 16, 16, 48), (64, 8, 20, 8), (5, 3, 3, 5)])
def test_copy_ld_and_accumulate(npix, c, sld, dld):
    L = pkg("_lib").lib()
    rng = np.random.RandomState(npix)
    src = rng.standard_normal((npix, sld)).astype(np.float32)
    dst0 = rng.standard_normal((npix, dld)).astype(np.float32)
    st = torch.cuda.current_stream().cuda_stream
    so, do = sld - c, dld - c
    dst, srcd = _t(dst0), _t(src)
    L.copy_ld(srcd.data_ptr() + 4 * so, sld, dst.data_ptr() + 4 * do, dld, npix, c, 0, st)
    ref = dst0.copy()
    ref[:, do:] = src[:, so:]
    assert np.array_equal(dst.cpu().numpy(), ref)
    L.copy_ld(srcd.data_ptr() + 4 * so, sld, dst.data_ptr() + 4 * do, dld, npix, c, 1, st)
    ref[:, do:] += src[:, so:]
    assert np.array_equal(dst.cpu().numpy(), ref)


@pytest.mark.parametrize("n,hw,c", [(2, 35, 16), (3, 16, 6), (1, 7, 3)])
def test_scale_nc_spatial_dropout_table(n, hw, c):
    """SpatialDropout2D: y[b, p, ch] = x[b, p, ch] * table[b, ch]."""
    L = pkg("_lib").lib()
    rng = np.random.RandomState(c)
    x = rng.standard_normal((n, hw, c)).astype(np.float32)
    tab = (rng.uniform(size=(n, c)) > 0.3).astype(np.float32) / 0.7
    y = torch.empty(n, hw, c, device=DEV)
    xd, td = _t(x), _t(tab)
    L.scale_nc(xd.data_ptr(), td.data_ptr(), y.data_ptr(), n, hw, c, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(y.cpu().numpy(), x * tab[:, None, :])


@pytest.mark.parametrize("case", [c for c in CONV_CASES] + [(1, 1, 16, 64, 1, 1, "valid", False, 3, 37, 41),
                                                             (3, 3, 16, 16, 1, 1, "same", True, 2, 50, 70),
                                                             (1, 1, 64, 256, 1, 1, "valid", True, 1, 20, 24)],
                         ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}".format(*c[:7]))
def test_conv2d_fwd_with_fused_bn_statistics(case):
    """The fused entry point must give the convolution output of ssmi_conv2d_fwd and the statistics
    ssmi_bn_train_stats computes from it (oracle: biased batch variance, Keras moving-average update)."""
    kh, kw, cin, cout, stride, dil, padding, bias, N, H, W = case
    eng = pkg("engine")
    L = pkg("_lib").lib()
    rng = np.random.RandomState(kh * 100 + cin + cout)
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=stride, dil=dil, padding=padding, bias=bias)
    store.materialise(DEV)
    ho, wo = cv.plan(N, H, W)
    w = rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)
    b = rng.standard_normal(cout) if bias else None
    x = rng.standard_normal((N, H, W, cin)) + 0.3
    store.views["c/kernel"].copy_(_t(w))
    if bias:
        store.views["c/bias"].copy_(_t(b))
    xd = _t(x)
    y = torch.empty(N, ho, wo, cout, device=DEV)
    mean, invstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    mm, mv = torch.full((cout,), 0.25, device=DEV), torch.full((cout,), 2.0, device=DEV)
    ws = torch.empty(cv.ws_bytes // 4 + 16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.conv2d_fwd_bn_stats(C.byref(cv.desc), xd.data_ptr(), cv.w.data_ptr(), cv.b.data_ptr() if bias else None, y.data_ptr(),
                          1e-3, 0.1, 0, mm.data_ptr(), mv.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
                          ws.numel() * 4, st)
    ref = K.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b) if bias else None, stride, dil, padding)
    _check(y, ref, "y")
    flat = ref.reshape(-1, cout).numpy()
    rm, rv = flat.mean(0), flat.var(0)
    _check(mean, rm, "mean", rtol=1e-4, atol_rel=1e-5)
    _check(invstd, 1.0 / np.sqrt(rv + 1e-3), "invstd", rtol=1e-4, atol_rel=1e-5)
    _check(mm, 0.1 * 0.25 + 0.9 * rm, "moving mean", rtol=1e-4, atol_rel=1e-5)
    _check(mv, 0.1 * 2.0 + 0.9 * rv, "moving variance", rtol=1e-4, atol_rel=1e-5)


@pytest.mark.parametrize("case", [
    # kh kw cin cout dil  N  H  W  act drop
    (1, 1, 16, 64, 1, 2, 9, 7, 1, False),      # conv3 of a bottleneck: 16x16 tiles, ragged pixels
    (3, 3, 16, 16, 1, 2, 10, 12, 1, False),
    (3, 3, 32, 32, 2, 1, 9, 9, 1, False),      # direct kernel, dilated
    (1, 5, 32, 32, 1, 1, 6, 11, 1, False),
    (1, 1, 32, 128, 1, 1, 8, 8, 0, False),
    (3, 3, 64, 40, 1, 2, 8, 8, 2, True),       # U-Net style: LeakyReLU + dropout table, 64 input channels
    (3, 3, 8, 24, 1, 3, 5, 6, 2, True),
    (1, 1, 4, 16, 1, 1, 16, 16, 1, False),
], ids=lambda c: "k{}x{}_{}to{}_d{}_act{}".format(c[0], c[1], c[2], c[3], c[4], c[8]))
def test_conv2d_bwd_data_with_fused_bn_backward(case):
    """ssmi_conv2d_bwd_data_bn_bwd == ssmi_conv2d_bwd_data followed by ssmi_bn_act_drop_bwd (checked against the oracle:
    autograd through BN(train) -> activation -> dropout -> convolution)."""
    kh, kw, cin, cout, dil, N, H, W, act, use_drop = case
    eng = pkg("engine")
    L = pkg("_lib").lib()
    rng = np.random.RandomState(kh * 10 + cin + cout + act)
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", kh, kw, cin, cout, dil=dil, padding="same", bias=False)
    bn = eng.BatchNorm(store, "bn", cin)
    if act == 1:
        bn.add_prelu(store, "pr")
    store.materialise(DEV)
    cv.plan(N, H, W)
    bn.plan(DEV)
    w = rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)
    xraw = rng.standard_normal((N, H, W, cin)) * 1.5 + 0.2
    gam, bet = rng.uniform(0.5, 1.5, cin), rng.standard_normal(cin) * 0.3
    alpha = rng.uniform(0.05, 0.3, cin)
    dy = rng.standard_normal((N, H, W, cout))
    drop = ((rng.uniform(size=(N, cin)) > 0.25) / 0.75) if use_drop else None
    # oracle
    xt = torch.tensor(xraw, requires_grad=True)
    gt, bt = torch.tensor(gam, requires_grad=True), torch.tensor(bet, requires_grad=True)
    at = torch.tensor(alpha, requires_grad=True)
    yb, mean, var = K.batch_norm_train(xt, gt, bt)
    a = K.prelu(yb, at) if act == 1 else (K.leaky_relu(yb, 0.1) if act == 2 else yb)
    if use_drop:
        a = K.spatial_dropout(a, torch.tensor(drop))
    a.retain_grad()
    out = K.conv2d(a, torch.tensor(w), None, 1, dil, "same")
    out.backward(torch.tensor(dy))
    # device
    store.views["c/kernel"].copy_(_t(w))
    bn.gamma.copy_(_t(gam)); bn.beta.copy_(_t(bet))
    if act == 1:
        bn.alpha.copy_(_t(alpha.reshape(1, 1, cin)))
    bn.mean.copy_(_t(mean.detach().numpy()))
    bn.invstd.copy_(_t(1.0 / np.sqrt(var.detach().numpy() + 1e-3)))
    xd, dyd = _t(xraw), _t(dy)
    dropd = _t(drop) if use_drop else None
    da = torch.empty(N, H, W, cin, device=DEV)
    dxbn = torch.empty_like(da)
    ws = torch.empty(max(cv.ws_bytes, L.bn_workspace_bytes(N * H * W, cin)) // 4 + 16, device=DEV)
    cv.bwd_data_bn(dyd, da, bn, xd, cin, dxbn, N * H * W, ws, torch.cuda.current_stream().cuda_stream,
                   act=act, slope=0.1, drop=dropd, hw=H * W)
    _check(da, a.grad, "da")
    _check(dxbn, xt.grad, "dx_bn")
    _check(bn.dgamma, gt.grad, "dgamma")
    _check(bn.dbeta, bt.grad, "dbeta")
    if act == 1:
        _check(bn.dalpha.reshape(-1), at.grad, "dalpha")


@pytest.mark.parametrize("variant", ["lds", "direct", "tile"])
def test_forced_conv_kernel_families(variant):
    """SSMI_CONV_KERNEL forces one convolution kernel family (read once per process): re-run the convolution operator
    tests in a child process for each, so the variants the selection rule does not pick stay covered."""
    import os
    import subprocess
    import sys
    if os.environ.get("SSMI_CONV_KERNEL"):
        pytest.skip("already inside a forced-variant run")
    env = dict(os.environ, SSMI_CONV_KERNEL=variant)
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_conv2d_fwd_bwd or fused_bn"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("case", [(1, 1, 128, 32, 1, 1, "valid", False, 2, 9, 7), (3, 3, 32, 32, 1, 2, "same", True, 1, 11, 13),
                                  (3, 3, 16, 16, 1, 1, "same", True, 2, 10, 12), (1, 1, 64, 16, 1, 1, "valid", False, 1, 33, 20),
                                  (3, 3, 64, 128, 1, 1, "same", True, 1, 8, 8), (2, 2, 16, 16, 2, 1, "valid", False, 2, 12, 8),
                                  (3, 3, 3, 13, 2, 1, "same", True, 1, 16, 16)],
                         ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}".format(*c[:7]))
def test_conv2d_fwd_bn_act_inference_fusion(case, act):
    """Inference phase: convolution + BN with fixed statistics + activation in one launch against
    oracle conv -> batch_norm_infer -> PReLU / LeakyReLU (both kernel families, ragged tiles, every activation)."""
    kh, kw, cin, cout, stride, dil, padding, bias, N, H, W = case
    eng = pkg("engine")
    L = pkg("_lib").lib()
    rng = np.random.RandomState(kh * 10 + cin + cout + act)
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=stride, dil=dil, padding=padding, bias=bias)
    store.materialise(DEV)
    ho, wo = cv.plan(N, H, W)
    w = rng.standard_normal((kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)
    b = rng.standard_normal(cout) if bias else None
    x = rng.standard_normal((N, H, W, cin))
    gamma, beta = rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.3
    mm, mv = rng.standard_normal(cout) * 0.2, rng.uniform(0.5, 1.5, cout)
    alpha = rng.uniform(0.05, 0.4, cout)
    store.views["c/kernel"].copy_(_t(w))
    if bias:
        store.views["c/bias"].copy_(_t(b))
    xd, gd, bd, ad = _t(x), _t(gamma), _t(beta), _t(alpha)
    mean, invstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    mmd, mvd = _t(mm), _t(mv)
    st = torch.cuda.current_stream().cuda_stream
    L.bn_infer_stats(mmd.data_ptr(), mvd.data_ptr(), cout, 1e-3, mean.data_ptr(), invstd.data_ptr(), st)
    y = torch.empty(N, ho, wo, cout, device=DEV)
    L.conv2d_fwd_bn_act_infer(C.byref(cv.desc), xd.data_ptr(), cv.w.data_ptr(), cv.b.data_ptr() if bias else None,
                              mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), act,
                              ad.data_ptr() if act == 1 else None, 0.1, y.data_ptr(), st)
    ref = K.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b) if bias else None, stride, dil, padding)
    ref = K.batch_norm_infer(ref, torch.from_numpy(gamma), torch.from_numpy(beta), torch.from_numpy(mm), torch.from_numpy(mv))
    if act == 1:
        ref = K.prelu(ref, torch.from_numpy(alpha))
    elif act == 2:
        ref = K.leaky_relu(ref, 0.1)
    _check(y, ref, "y")
    # and it is the composition of the two separate launches bit for bit (the BN kernel takes channel counts % 4 == 0)
    if cout % 4:
        return
    y0 = torch.empty_like(y)
    y1 = torch.empty_like(y)
    cv.fwd(xd, y0, st)
    L.bn_act_fwd(y0.data_ptr(), cout, mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), act,
                 ad.data_ptr() if act == 1 else None, 0.1, y1.data_ptr(), N * ho * wo, cout, st)
    assert torch.equal(y, y1)


@pytest.mark.parametrize("cin,cout,N,H,W", [(32, 128, 2, 9, 7), (16, 64, 1, 16, 24), (4, 16, 2, 10, 12), (8, 40, 1, 5, 27),
                                            (32, 128, 1, 32, 32)])
def test_conv2d_fwd_bn_add_act_inference_fusion(cin, cout, N, H, W):
    """Inference-phase tail of a regular bottleneck in one launch, PReLU(BN3(conv3(a2)) + x), against the oracle and, bit
    for bit, against the two launches it replaces (ssmi_conv2d_fwd -> ssmi_merge_fwd with the identity skip)."""
    eng, lib_mod = pkg("engine"), pkg("_lib")
    L = lib_mod.lib()
    rng = np.random.RandomState(cin + cout + H)
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", 1, 1, cin, cout, padding="valid", bias=False)
    store.materialise(DEV)
    cv.plan(N, H, W)
    w = rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)
    x = rng.standard_normal((N, H, W, cin))
    skip = rng.standard_normal((N, H, W, cout))
    gamma, beta = rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.3
    mm, mv = rng.standard_normal(cout) * 0.2, rng.uniform(0.5, 1.5, cout)
    alpha = rng.uniform(0.05, 0.4, cout)
    store.views["c/kernel"].copy_(_t(w))
    xd, sd, gd, bd, ad, mmd, mvd = (_t(v) for v in (x, skip, gamma, beta, alpha, mm, mv))
    mean, invstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.bn_infer_stats(mmd.data_ptr(), mvd.data_ptr(), cout, 1e-3, mean.data_ptr(), invstd.data_ptr(), st)
    y = torch.empty(N, H, W, cout, device=DEV)
    L.conv2d_fwd_bn_add_act_infer(C.byref(cv.desc), xd.data_ptr(), cv.w.data_ptr(), None, mean.data_ptr(), invstd.data_ptr(),
                                  gd.data_ptr(), bd.data_ptr(), sd.data_ptr(), cout, 1, ad.data_ptr(), 0.0, y.data_ptr(), st)
    ref = K.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 1, 1, "valid")
    ref = K.batch_norm_infer(ref, torch.from_numpy(gamma), torch.from_numpy(beta), torch.from_numpy(mm), torch.from_numpy(mv))
    ref = K.prelu(ref + torch.from_numpy(skip), torch.from_numpy(alpha))
    _check(y, ref, "y")
    y3, out = torch.empty_like(y), torch.empty_like(y)
    cv.fwd(xd, y3, st)
    md = lib_mod.MergeDesc(N, H, W, cout, 0, cout)
    bn3 = lib_mod.BnParams(mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr())
    L.merge_fwd(C.byref(md), y3.data_ptr(), C.byref(bn3), None, sd.data_ptr(), None, None, ad.data_ptr(), out.data_ptr(), st)
    assert torch.equal(y, out)
    with pytest.raises(lib_mod.SsmiError):      # a skip tensor narrower than the output
        L.conv2d_fwd_bn_add_act_infer(C.byref(cv.desc), xd.data_ptr(), cv.w.data_ptr(), None, mean.data_ptr(),
                                      invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), sd.data_ptr(), cout - 4, 1,
                                      ad.data_ptr(), 0.0, y.data_ptr(), st)


def _fuzz_cases(n, seed):
    """Seeded random convolution configurations: every combination of kernel extent, stride, dilation, padding mode,
    channel counts around the tile widths (4 / 16 / 32 / 64) and ragged spatial sizes."""
    rng = np.random.RandomState(seed)
    chans = [2, 3, 4, 5, 8, 12, 13, 16, 17, 24, 31, 32, 33, 40, 48, 64, 68, 96, 128]
    out = []
    while len(out) < n:
        kh, kw = [(1, 1), (2, 2), (3, 3), (1, 5), (5, 1), (3, 1), (1, 3)][rng.randint(7)]
        s = int(rng.choice([1, 1, 2]))
        d = 1 if s == 2 else int(rng.choice([1, 1, 2, 3]))
        pad = "valid" if (kh, kw) in ((1, 1), (2, 2)) else str(rng.choice(["same", "same", "valid"]))
        cin, cout = int(rng.choice(chans)), int(rng.choice(chans))
        N, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 24)), int(rng.randint(1, 40))
        eh, ew = (kh - 1) * d + 1, (kw - 1) * d + 1
        if pad == "valid" and (H < eh or W < ew):
            continue
        if (kh, kw) == (2, 2) and s == 2 and (H % 2 or W % 2):
            H, W = H + H % 2, W + W % 2
        out.append((kh, kw, cin, cout, s, d, pad, bool(rng.randint(2)), N, H, W))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(160, 20240611), ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}_b{}_{}x{}x{}".format(*c))
def test_conv2d_fuzz(case):
    """Forward, data gradient (with and without the fused addend) and weight / bias gradient of random configurations
    against the fp64 oracle."""
    test_conv2d_fwd_bwd(case)


def _fuzz_bn_bwd_cases(n, seed):
    """(kh kw cin cout dil N H W act drop) for the fused data-gradient + BN-backward entry point (stride 1, channel
    counts a multiple of 4 as the BN kernels require)."""
    rng = np.random.RandomState(seed)
    chans = [4, 8, 12, 16, 24, 32, 40, 64, 128]
    out = []
    for _ in range(n):
        kh, kw = [(1, 1), (3, 3), (1, 5), (5, 1)][rng.randint(4)]
        out.append((kh, kw, int(rng.choice(chans)), int(rng.choice(chans)), int(rng.choice([1, 1, 2, 4])), int(rng.randint(1, 4)),
                    int(rng.randint(2, 20)), int(rng.randint(2, 36)), int(rng.randint(3)), bool(rng.randint(2))))
    return out


@pytest.mark.parametrize("case", _fuzz_bn_bwd_cases(50, 77), ids=lambda c: "k{}x{}_{}to{}_d{}_{}x{}x{}_a{}_drop{}".format(*c))
def test_conv2d_bwd_data_bn_bwd_fuzz(case):
    test_conv2d_bwd_data_with_fused_bn_backward(case)


@pytest.mark.parametrize("case", _fuzz_cases(50, 99),
                         ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}_b{}_{}x{}x{}".format(*c))
def test_conv2d_fwd_bn_stats_fuzz(case):
    test_conv2d_fwd_with_fused_bn_statistics(case)


def _rand_cases(n, seed, gen):
    rng = np.random.RandomState(seed)
    return [gen(rng) for _ in range(n)]


@pytest.mark.parametrize("c,npix,act", _rand_cases(30, 5, lambda r: (int(r.choice([4, 8, 12, 16, 24, 32, 64, 100, 128, 256])),
                                                                      int(r.randint(1, 3000)), int(r.randint(3)))))
def test_bn_act_fuzz(c, npix, act):
    test_bn_act_fwd_bwd(c, npix, act)


@pytest.mark.parametrize("n,h,w,c", _rand_cases(30, 6, lambda r: (int(r.randint(1, 4)), int(r.randint(1, 20)), int(r.randint(1, 30)),
                                                                   int(r.choice([1, 3, 4, 5, 13, 16, 24, 64, 128])))))
def test_maxpool_fuzz(n, h, w, c):
    test_maxpool_argmax_unpool_bit_exact(n, h, w, c)


@pytest.mark.parametrize("C_,npix", _rand_cases(20, 7, lambda r: (int(r.randint(2, 65)), int(r.randint(1, 4000)))))
def test_softmax_xent_fuzz(C_, npix):
    test_softmax_xent_loss_grad_cm(C_, npix)


def test_device_timestamp_orders_stream_work():
    """ssmi_device_timestamp: stamps taken on one stream are ordered, the rate is reported, NULL is rejected."""
    import time
    libm = pkg("_lib")
    L = libm.lib()
    t = torch.zeros(2, dtype=torch.int64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.device_timestamp(t.data_ptr(), st)
    torch.cuda.synchronize()
    time.sleep(0.02)
    L.device_timestamp(t.data_ptr() + 8, st)
    torch.cuda.synchronize()
    khz = L.device_timestamp_khz()
    assert khz > 0
    ms = float(t[1] - t[0]) / khz
    assert 15.0 < ms < 2000.0, ms
    with pytest.raises(libm.SsmiError):
        L.device_timestamp(None, st)
