"""Encoder-only classification models on the GPU (src/models.py:985-993, 1040-1048, 1346-1354, 1665-1668 and the
classification lambda loss src/losses.py:579-652) against the CPU oracle: rtol 1e-3 for loss, logits and gradients."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E
from oracle import tf_keras_ops as K

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def _close(got, ref, what, rtol=1e-3, atol_rel=1e-4):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    err = np.abs(got - ref)
    bad = err > atol_rel * np.abs(ref).max() + rtol * np.abs(ref)
    assert not bad.any(), "{}: {} of {} off, max err {:.3e}".format(what, int(bad.sum()), ref.size, float(err.max()))


@pytest.mark.parametrize("n,hw,c", [(2, 64, 128), (3, 5, 24), (1, 1, 7), (4, 4096, 512), (2, 9, 70)])
def test_global_average_pool(n, hw, c):
    L = pkg("_lib").lib()
    rng = np.random.RandomState(hw)
    x = rng.standard_normal((n, hw, c)).astype(np.float32)
    dy = rng.standard_normal((n, c)).astype(np.float32)
    xd, dyd = _t(x), _t(dy)
    y = torch.empty(n, c, device=DEV)
    dx = torch.empty(n, hw, c, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    L.global_avgpool_fwd(xd.data_ptr(), n, hw, c, y.data_ptr(), st)
    L.global_avgpool_bwd(dyd.data_ptr(), n, hw, c, dx.data_ptr(), st)
    _close(y, x.astype(np.float64).mean(axis=1), "mean", rtol=1e-5, atol_rel=1e-6)
    assert np.array_equal(dx.cpu().numpy(), np.broadcast_to((dy / np.float32(hw))[:, None, :], (n, hw, c)))


@pytest.mark.parametrize("n,C", [(5, 24), (300, 7), (1, 2)])
def test_classification_loss_and_gradient(n, C):
    losses = pkg("losses")
    rng = np.random.RandomState(n)
    z = rng.standard_normal((n, C)) * 4
    z[0, C - 1] = -40.0                                   # softmax below the clip: term -log(1e-6), no gradient
    y = (rng.uniform(size=(n, C)) < 0.3).astype(np.float64)   # multi-hot labels
    y[0, C - 1] = 1.0
    w = rng.uniform(0.2, 2.0, size=(n, C))
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    ref = K.classification_weighted_categorical_crossentropy(torch.tensor(y), zt, torch.tensor(w))
    (g,) = torch.autograd.grad(ref, [zt])
    loss = losses.classification_categorical_crossentropy_lambda_loss(
        [_t(z), _t(y), _t(w), torch.zeros(n, 1, dtype=torch.int32, device=DEV)])
    _close(loss, ref.item(), "loss")
    _close(losses.last_gradient(), g.numpy(), "dlogits")
    cw = rng.uniform(0.5, 1.5, size=C)
    ref2 = K.classification_weighted_categorical_crossentropy(torch.tensor(y), zt, torch.tensor(cw))
    _close(losses.classification_weighted_categorical_crossentropy_loss(cw)(_t(y), _t(z)), ref2.item(), "class-weight form")
    with pytest.raises(ValueError):
        losses.classification_categorical_crossentropy_lambda_loss([_t(z), _t(y), _t(w), torch.ones(n, 1, dtype=torch.int32, device=DEV)])


@pytest.mark.parametrize("name,B,H,W", [("enet-naive-upsampling-encoder-only", 3, 64, 64), ("enet-max-unpooling-encoder-only", 2, 64, 96),
                                        ("enet-naive-upsampling-enhanced-encoder-only", 2, 32, 32), ("segnet-encoder-only", 2, 32, 32)])
def test_encoder_only_training_step(name, B, H, W, monkeypatch):
    """One training step with smooth activations (every PReLU alpha 1 / LeakyReLU slope 1): logits, loss and every weight
    gradient against the fp64 oracle, eagerly and as replayed HIP graphs; Adam-updated weights stay finite."""
    models, losses, opt = pkg("models"), pkg("losses"), pkg("keras_extensions.extended_optimizers")
    nc, seed = 24, 17
    w0 = E.init_weights(name, nc, seed=seed, perturb=True)
    for k in w0:
        if k.endswith("/alpha"):
            w0[k] = np.ones_like(w0[k])
    if name.startswith("segnet"):
        monkeypatch.setattr(E.V, "LEAKY_ALPHA", 1.0)
        monkeypatch.setattr(pkg("engine_unet"), "LEAKY_ALPHA", 1.0)
    mb = models.get_model(name, (H, W, 3), nc, losses.ModelLambdaLossType.CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY)
    model = mb.model
    assert model.weight_names() == list(w0.keys())
    model.compile(optimizer=opt.Adam(lr=5e-4), loss={'loss': losses.dummy_loss, 'logits': lambda t, p: 0 * p},
                  loss_weights={'loss': 1.0, 'logits': 0.0})
    rng = np.random.RandomState(seed)
    images = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    y = (rng.uniform(size=(B, nc)) < 0.2).astype(np.float32)
    cw = np.broadcast_to(rng.uniform(0.5, 1.5, size=nc).astype(np.float32), (B, nc)).copy()
    masks = E.make_dropout_masks(name, B, seed)
    ref = E.train_step(name, w0, images, y, cw, masks, collect=True, classification=True)
    stream = torch.cuda.Stream()
    for step in range(2):
        model.set_weights([w0[k] for k in model.weight_names()])
        model._ensure_buffers(B, H, W)
        g = model.graph
        saved = g.drop_groups
        g.drop_groups = []
        for k, m in masks.items():
            g.drop_views[k].copy_(torch.from_numpy(m.astype(np.float32)))
        model.use_hip_graph = step > 0
        with torch.cuda.stream(stream):
            outs = model.train_on_batch([images, y, cw, np.zeros((B, 1), np.int32)], None)
        torch.cuda.synchronize()
        g.drop_groups = saved
        assert tuple(g.logits.shape) == (B, nc)
        _close(g.logits, ref["logits"], "logits")
        _close(outs[0], ref["loss"], "loss", atol_rel=1e-6)
        _close(g.dlogits, ref["dlogits"], "dlogits")
        for k, gr in ref["grads"].items():
            got = g.store.gviews[k].cpu().numpy().astype(np.float64)
            scale = np.abs(gr).max()
            assert np.abs(got - gr).max() <= 1e-3 * scale + 2e-6, "grad {} (step {})".format(k, step)
        assert all(np.isfinite(v).all() for v in model.get_weights())
    pred = model.predict_on_batch(images)
    assert pred.shape == (B, nc)
