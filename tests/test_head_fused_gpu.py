"""The ENets' last layer inside the loss kernel (ssmi_logits_tconv2x2_xent_fwd_bwd, through the C ABI): logits =
Conv2DTranspose(24, (2, 2), strides 2, 'same') (src/models.py:1318) + the weighted pixelwise cross-entropy
(src/losses.py:243-265) + the streaming confusion matrix (src/metrics.py:355-378) in one launch.

Checked against (1) the CPU oracle (float64 transposed convolution + loss), tolerance 1e-3 as for every float32 operator,
and (2) the two-call route of the same library: the logits the fused kernel stores, fed to ssmi_softmax_xent_fwd_bwd, must
give the SAME bits for gradient, matrix and predictions (the arithmetic behind the logits is one piece of code) and the
same loss up to the order of its summation over pixels, and must agree with ssmi_conv2d_transpose_fwd to float32 / bfloat16 rounding."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"
NC, CIN = 24, 16


def _run(dtype, N, H, W, npix_metrics_frac, with_outputs, seed):
    from oracle import tf_keras_ops as K
    eng, libm = pkg("engine"), pkg("_lib")
    L = libm.lib()
    bf = dtype == "bf16"
    adt = torch.bfloat16 if bf else torch.float32
    code = libm.BF16 if bf else libm.F32
    LT = libm.lib_for(code)
    rng = np.random.RandomState(seed)
    store = eng.ParamStore(adt)
    cv = eng.Conv(store, "logits", 2, 2, CIN, NC, stride=2, bias=True, transpose=True)
    store.materialise(torch.device(DEV))
    w = rng.standard_normal((2, 2, NC, CIN)).astype(np.float32) * 0.5
    b = rng.standard_normal(NC).astype(np.float32)
    store.views["logits/kernel"].copy_(torch.from_numpy(w).to(DEV))
    store.views["logits/bias"].copy_(torch.from_numpy(b).to(DEV))
    assert cv.plan(N, H, W) == (2 * H, 2 * W)
    x = rng.standard_normal((N, H, W, CIN)).astype(np.float32)
    xd = torch.from_numpy(x).to(DEV).to(adt)
    x_used = xd.float().cpu().numpy()          # what the kernels see (bf16-rounded for bf16 storage)
    npix = N * 4 * H * W
    labels = rng.randint(0, NC, size=(N, 2 * H, 2 * W)).astype(np.int32)
    labels.reshape(-1)[rng.randint(0, npix, size=3)] = NC + 5        # out-of-range labels: counted, contribute nothing
    pw = rng.uniform(0.0, 2.0, size=(N, 2 * H, 2 * W)).astype(np.float32)
    npix_m = int(npix * npix_metrics_frac)
    ignore_bits = 1          # class 0 ignored by the matrix
    ld, wd = torch.from_numpy(labels).to(DEV), torch.from_numpy(pw).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(L.softmax_xent_workspace_bytes(npix) // 4 + 8, device=DEV)

    def outputs():
        return dict(logits=torch.full((N, 2 * H, 2 * W, NC), 7.0, device=DEV).to(adt),
                    dlogits=torch.full((N, 2 * H, 2 * W, NC), 7.0, device=DEV).to(adt),
                    loss=torch.zeros(1, device=DEV), cm=torch.zeros(NC, NC, dtype=torch.int64, device=DEV),
                    pred=torch.full((npix,), -1, dtype=torch.int32, device=DEV), bad=torch.zeros(1, dtype=torch.int32, device=DEV))

    f = outputs()
    assert L.logits_tconv2x2_xent_supported(CIN, NC)
    L.logits_tconv2x2_xent_fwd_bwd(code, xd.data_ptr(), N, H, W, CIN, CIN, store.views["logits/kernel"].data_ptr(),
                                   store.views["logits/bias"].data_ptr(), NC, ld.data_ptr(), wd.data_ptr(), npix, 1.0,
                                   f["logits"].data_ptr() if with_outputs else None,
                                   f["dlogits"].data_ptr() if with_outputs else None, f["loss"].data_ptr(),
                                   f["cm"].data_ptr(), npix_m, ignore_bits, f["pred"].data_ptr(), f["bad"].data_ptr(),
                                   ws.data_ptr(), ws.numel() * 4, st)
    torch.cuda.synchronize()
    assert int(f["bad"].item()) == len(np.unique(np.flatnonzero(labels.reshape(-1) >= NC)))

    # (1) the oracle on the tensors the kernel saw
    zr = K.conv2d_transpose(torch.tensor(x_used, dtype=torch.float64), torch.tensor(w, dtype=torch.float64),
                            torch.tensor(b, dtype=torch.float64), 2).numpy()
    tol = 2e-2 if bf else 1e-3
    if with_outputs:
        z = f["logits"].float().cpu().numpy()
        assert np.abs(z - zr).max() <= tol * np.abs(zr).max()
    else:
        assert float(f["logits"].float().min()) == 7.0 and float(f["dlogits"].float().min()) == 7.0, "optional outputs were written"
    valid = labels.reshape(-1) < NC
    lse = np.log(np.exp(zr - zr.max(-1, keepdims=True)).sum(-1)) + zr.max(-1)
    zl = np.take_along_axis(zr.reshape(-1, NC), np.where(valid, labels.reshape(-1), 0)[:, None], 1)[:, 0]
    loss_ref = float((np.where(valid, pw.reshape(-1), 0.0) * (lse.reshape(-1) - zl)).sum() / npix)
    assert abs(float(f["loss"].item()) - loss_ref) <= tol * abs(loss_ref)

    # (2) the two-call route on the fused kernel's own logits: identical bits
    if with_outputs:
        t = outputs()
        LT.softmax_xent_fwd_bwd(f["logits"].data_ptr(), ld.data_ptr(), wd.data_ptr(), npix, NC, 1.0, t["dlogits"].data_ptr(),
                                t["loss"].data_ptr(), t["cm"].data_ptr(), npix_m, ignore_bits, t["pred"].data_ptr(),
                                t["bad"].data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
        torch.cuda.synchronize()
        # (the loss itself is a sum over pixels taken in a different order: equal to float32 summation error)
        assert abs(float(t["loss"].item()) - float(f["loss"].item())) <= 2e-6 * abs(float(t["loss"].item()))
        assert torch.equal(t["dlogits"].view(torch.int16 if bf else torch.int32), f["dlogits"].view(torch.int16 if bf else torch.int32))
        assert torch.equal(t["cm"], f["cm"]) and torch.equal(t["pred"], f["pred"])
        assert int(f["cm"].sum().item()) == int(((labels.reshape(-1)[:npix_m] < NC) & (labels.reshape(-1)[:npix_m] != 0)).sum())
        # ... and the library's own transposed convolution gives the same logits up to rounding
        y = torch.empty(N, 2 * H, 2 * W, NC, device=DEV, dtype=adt)
        cv.fwd(xd, y, st)
        torch.cuda.synchronize()
        d = (y.float() - f["logits"].float()).abs().max().item()
        assert d <= (2.0 ** -7 if bf else 1e-5) * np.abs(zr).max()
    return f


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,H,W,frac", [(2, 5, 7, 1.0), (1, 3, 40, 0.5), (3, 16, 16, 1.0)])
def test_fused_head_matches_oracle_and_two_call_route(dtype, N, H, W, frac):
    _run(dtype, N, H, W, frac, True, seed=N * 100 + H)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_head_without_optional_outputs(dtype):
    a = _run(dtype, 2, 6, 9, 1.0, True, seed=5)
    b = _run(dtype, 2, 6, 9, 1.0, False, seed=5)
    assert torch.equal(a["loss"], b["loss"]) and torch.equal(a["cm"], b["cm"]) and torch.equal(a["pred"], b["pred"])


def test_unsupported_shapes_are_rejected():
    libm = pkg("_lib")
    L = libm.lib()
    assert not L.logits_tconv2x2_xent_supported(16, 12) and not L.logits_tconv2x2_xent_supported(4, 24)
    t = torch.zeros(64, device=DEV)
    with pytest.raises(libm.SsmiError):
        L.logits_tconv2x2_xent_fwd_bwd(libm.F32, t.data_ptr(), 1, 1, 1, 16, 16, t.data_ptr(), t.data_ptr(), 12, t.data_ptr(),
                                       t.data_ptr(), 4, 1.0, None, None, t.data_ptr(), None, 0, 0, None, None, t.data_ptr(), 256,
                                       torch.cuda.current_stream().cuda_stream)


def test_step_without_stored_logits_equals_step_with_them():
    """The default step (logits never stored) and the keep_logits step are the same kernel: identical loss, matrix and
    updated weights; reading `graph.logits` after the former is an error instead of stale data."""
    from oracle import enet_oracle as E
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    name, B, H, W = "enet-naive-upsampling", 2, 32, 32
    w0 = E.init_weights(name, NC, seed=3, perturb=True)
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=4, block=8)
    res = []
    for keep in (False, True):
        m = models.get_model(name, (None, None, 3), NC, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY).model
        m.set_weights([w0[k] for k in m.weight_names()])
        m.compile(optimizer=xo.Adam(lr=1e-3), metrics={'logits': [metrics.segmentation_confusion_matrix(NC, 0, [0])]})
        m.keep_logits = keep
        with torch.cuda.stream(torch.cuda.Stream()):
            outs = [m.train_on_batch([images, labels, pw, np.zeros((B, 1), np.int32)]) for _ in range(2)]
            ev = m.test_on_batch([images, labels, pw, np.zeros((B, 1), np.int32)])
        torch.cuda.synchronize()
        assert m.graph.head_input is not None, "the step did not take the fused route"
        if keep:
            assert m.graph.logits.shape == (B, H, W, NC)
        else:
            with pytest.raises(RuntimeError):
                m.graph.logits
        res.append((outs, ev, [w.copy() for w in m.get_weights()]))
    (o0, e0, w0_), (o1, e1, w1_) = res
    for a, b in zip(o0 + [e0], o1 + [e1]):
        for u, v in zip(a, b):
            assert np.array_equal(np.asarray(u), np.asarray(v))
    for u, v in zip(w0_, w1_):
        assert np.array_equal(u, v)
