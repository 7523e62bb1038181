"""GPU parity: one full training step (forward, loss, backward, BN moving updates, Adam) of the HIP path against the
CPU oracle (fp64) on identical weights, inputs and dropout masks.  Tolerances: rtol 1e-3 for fp32 activations, loss
and gradients (BASELINE.json north_star); pooling indices and confusion matrices bit-exact."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), "{}: {} of {} elements off; max abs err {:.3e} (ref scale {:.3e})".format(
        what, int(bad.sum()), a.size, float(err.max()), float(np.abs(b).max()))


def _run_case(model_name, B, H, W, seed, steps=2):
    models = pkg("models")
    losses = pkg("losses")
    metrics = pkg("metrics")
    opt = pkg("keras_extensions.extended_optimizers")
    nc = 24
    w0 = E.init_weights(model_name, nc, seed=seed, perturb=True)
    mb = models.get_model(model_name, (None, None, 3), nc, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY)
    model = mb.model
    assert model.weight_names() == list(w0.keys())
    model.set_weights([w0[k] for k in model.weight_names()])
    adam = dict(lr=5e-4, beta_1=0.9, beta_2=0.999, eps=1e-8)
    model.compile(optimizer=opt.Adam(lr=5e-4), loss={'loss': losses.dummy_loss, 'logits': lambda t, p: 0 * p},
                  loss_weights={'loss': 1.0, 'logits': 0.0},
                  metrics={'logits': [metrics.segmentation_accuracy(0, [0]), metrics.segmentation_mean_iou(nc, 0, [0]),
                                      metrics.segmentation_mean_per_class_accuracy(nc, 0, [0]),
                                      metrics.segmentation_confusion_matrix(nc, 0, [0])]})
    cw = np.ones(nc, dtype=np.float32)
    cw[0] = 0
    cw[1:] = 0.5 + np.arange(1, nc) / 23.0
    weights = w0
    state = {"iterations": 0, "m": {}, "v": {}}
    cm_total = np.zeros((nc, nc), dtype=np.int64)
    stream = torch.cuda.Stream()
    for step in range(steps):
        images, labels, pw = E.synthetic_batch(B, H, W, nc, seed=seed + 100 + step, block=8, class_weights=cw)
        masks = E.make_dropout_masks(model_name, B, seed + step)
        # explicit masks on the device (the RNG is bypassed by writing the tables directly)
        model._ensure_buffers(B, H, W)
        g = model.graph
        saved_groups = g.drop_groups
        g.drop_groups = []
        for name, m in masks.items():
            g.drop_views[name].copy_(torch.from_numpy(m.astype(np.float32)))
        ref = E.train_step(model_name, weights, images, labels, pw, masks, adam=adam, adam_state=state, collect=True)
        with torch.cuda.stream(stream):
            # fetch pre-update quantities with an eager run of the same segments
            model.use_hip_graph = (step > 0)
            outs = model.train_on_batch([images, labels, pw, np.zeros((B, 1), dtype=np.int32)],
                                        [np.zeros(B, dtype=np.int32), labels[..., None]])
        torch.cuda.synchronize()
        g.drop_groups = saved_groups
        cm_total += ref["cm"]
        # loss / logits / dlogits
        _close(outs[0], ref["loss"], 1e-3, 1e-6, "loss step {}".format(step))
        _close(g.logits.cpu().numpy(), ref["logits"], 1e-3, 1e-3 * np.abs(ref["logits"]).max() * 1e-1, "logits")
        _close(g.dlogits.cpu().numpy(), ref["dlogits"], 1e-3, 1e-3 * np.abs(ref["dlogits"]).max(), "dlogits")
        # pooling indices bit-exact
        for bname in ("en_bn_1.0", "en_bn_2.0"):
            key = bname + "_other_pool2d/argmax"
            if key in ref["acts"]:
                got = g.by_name[bname].idx.cpu().numpy().astype(np.int64)
                assert np.array_equal(got, ref["acts"][key]), "argmax indices of {} differ".format(bname)
        # gradients: per tensor, tolerance relative to the tensor's own scale
        gv = g.store.gviews
        for k, gr in ref["grads"].items():
            got = gv[k].cpu().numpy()
            scale = max(np.abs(gr).max(), 1e-12)
            _close(got, gr, 1e-3, 2e-3 * scale, "grad " + k)
        # confusion matrix bit-exact (cumulative)
        assert np.array_equal(outs[6], cm_total), "confusion matrix differs"
        m = metrics.metrics_from_confusion_matrix(cm_total, [0])
        assert abs(outs[3] - m["accuracy"]) < 1e-12
        # updated weights (Adam + BN moving statistics)
        new_w = model.get_weights_dict()
        for k, v in ref["new_weights"].items():
            scale = max(np.abs(v).max(), 1e-6)
            _close(new_w[k], v, 1e-3, 2e-3 * scale, "updated " + k)
        # continue from the ORACLE's weights so that errors do not compound across steps
        weights = ref["new_weights"]
        model.set_weights([weights[k] for k in model.weight_names()])


def test_enet_naive_train_step_parity():
    _run_case("enet-naive-upsampling", 2, 64, 64, seed=3, steps=2)


def test_enet_max_unpooling_train_step_parity():
    _run_case("enet-max-unpooling", 2, 64, 96, seed=5, steps=1)
