"""Semi-supervised losses on the GPU (mean teacher, superpixel, both): consistency cost + gradient, the lambda loss with unlabeled rows, the teacher's EMA
update and a whole student step, against the CPU oracle and the golden vectors.

Tolerances: rtol 1e-3 (fp32 kernels against fp64 oracle values) for loss and gradients; the EMA update is bit-exact
against numpy float32 arithmetic (two roundings of the products, one of the sum - no fused multiply-add); confusion
matrices bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E
from oracle import tf_keras_ops as K

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def _close(got, ref, what, rtol=1e-3, atol_rel=1e-4):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    err = np.abs(got - ref)
    bad = err > atol_rel * np.abs(ref).max() + rtol * np.abs(ref)
    assert not bad.any(), "{}: {} of {} off, max err {:.3e}".format(what, int(bad.sum()), ref.size, float(err.max()))


def test_lambda_loss_golden_vector():
    losses = pkg("losses")
    g = np.load(os.path.join(GOLD, "ops.npz"))
    args = [_t(g["mt_logits"]), _t(g["mt_labels"], torch.int32), _t(g["mt_weights"]),
            torch.ones(3, 1, dtype=torch.int32, device=DEV), _t(g["mt_teacher"]), torch.full((3, 1), 3.5, device=DEV)]
    loss = losses.segmentation_mean_teacher_lambda_loss(args)
    _close(loss, g["mt_loss"], "mean-teacher loss")
    _close(losses.last_gradient(), g["mt_dlogits"], "mean-teacher dlogits")
    with pytest.raises(ValueError):
        losses.segmentation_mean_teacher_lambda_loss(args[:4])


@pytest.mark.parametrize("C,npix_shape,nu", [(24, (4, 33, 20), 1), (24, (2, 16, 16), 0), (5, (3, 9, 7), 2), (64, (2, 5, 5), 1)])
def test_consistency_kernel_against_oracle(C, npix_shape, nu):
    """Both code paths (24 classes in registers, any other count re-reading), with and without rows to accumulate into."""
    losses = pkg("losses")
    rng = np.random.RandomState(C * 7 + nu)
    B = npix_shape[0]
    logits = rng.standard_normal(npix_shape + (C,)) * 3
    teacher = rng.standard_normal(npix_shape + (C,)) * 3
    labels = rng.randint(0, C, size=npix_shape)
    w = rng.uniform(0.2, 2.0, size=npix_shape)
    lg = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    ref = K.segmentation_mean_teacher_lambda_loss(lg, torch.tensor(labels), torch.tensor(w), nu, torch.tensor(teacher), 0.8)
    (dref,) = torch.autograd.grad(ref, [lg])
    args = [_t(logits), _t(labels, torch.int32), _t(w), torch.full((B, 1), nu, dtype=torch.int32, device=DEV), _t(teacher),
            torch.full((B, 1), 0.8, device=DEV)]
    loss = losses.segmentation_mean_teacher_lambda_loss(args)
    _close(loss, ref.item(), "loss")
    _close(losses.last_gradient(), dref.numpy(), "dlogits")


def test_xent_counts_metric_rows_beyond_the_loss_rows():
    """npix_metrics > npix: the unlabeled rows reach the confusion matrix, not the loss; their gradient rows stay untouched."""
    L = pkg("_lib").lib()
    rng = np.random.RandomState(3)
    C, n, hw = 24, 3, 40
    logits = rng.standard_normal((n, hw, C)).astype(np.float32)
    labels = rng.randint(0, C, size=(n, hw)).astype(np.int32)
    labels[2] = 0
    w = rng.uniform(0.5, 1.5, size=(n, hw)).astype(np.float32)
    tl, tlab, tw = _t(logits), _t(labels, torch.int32), _t(w)
    dl = torch.full((n, hw, C), 7.0, device=DEV)
    loss = torch.zeros(1, device=DEV)
    cm = torch.zeros(C, C, dtype=torch.int64, device=DEV)
    ws = torch.empty(L.softmax_xent_workspace_bytes(n * hw) // 4 + 4, device=DEV)
    L.softmax_xent_fwd_bwd(tl.data_ptr(), tlab.data_ptr(), tw.data_ptr(), 2 * hw, C, 1.0, dl.data_ptr(), loss.data_ptr(),
                           cm.data_ptr(), n * hw, 0, None, None, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(cm.cpu().numpy(), K.confusion_matrix_np(labels, logits, C, []))
    ref = K.sparse_weighted_pixelwise_xent(torch.tensor(logits[:2], dtype=torch.float64), torch.tensor(labels[:2].astype(np.int64)),
                                           torch.tensor(w[:2], dtype=torch.float64))
    _close(loss[0], ref.item(), "loss over the labeled rows")
    assert bool((dl[2] == 7.0).all()) and not bool((dl[:2] == 7.0).any())


def test_ema_update_bit_exact():
    L = pkg("_lib").lib()
    g = np.load(os.path.join(GOLD, "ops.npz"))
    for s in "ab":
        t = _t(g["ema_t_" + s])
        st = _t(g["ema_s_" + s])
        L.ema_update(t.data_ptr(), st.data_ptr(), 0.99, t.numel(), torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(t.cpu().numpy(), g["ema_out_" + s])
    rng = np.random.RandomState(0)
    for a in (0.0, 0.5, 0.999, 1.0):
        tw, sw = rng.standard_normal(100003).astype(np.float32), rng.standard_normal(100003).astype(np.float32)
        t, st = _t(tw), _t(sw)
        L.ema_update(t.data_ptr(), st.data_ptr(), a, t.numel(), torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(t.cpu().numpy(), K.ema_update_np([tw], [sw], a)[0]), a


def _compile(model, nc=24):
    losses, metrics, opt = pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    model.compile(optimizer=opt.Adam(lr=5e-4), loss={'loss': losses.dummy_loss, 'logits': lambda t, p: 0 * p},
                  loss_weights={'loss': 1.0, 'logits': 0.0},
                  metrics={'logits': [metrics.segmentation_accuracy(0, [0]), metrics.segmentation_mean_iou(nc, 0, [0]),
                                      metrics.segmentation_mean_per_class_accuracy(nc, 0, [0]),
                                      metrics.segmentation_confusion_matrix(nc, 0, [0])]})


def test_student_step_against_oracle():
    """A whole student step (ENet, smooth activations: every PReLU alpha 1) with one unlabeled row: loss, logits gradient,
    every weight gradient and the confusion matrix, eager and as replayed HIP graphs with a different coefficient."""
    models, losses = pkg("models"), pkg("losses")
    name, nc, B, H, W, seed = "enet-naive-upsampling", 24, 3, 32, 32, 11
    w0 = E.init_weights(name, nc, seed=seed, perturb=True)
    for k in w0:
        if k.endswith("/alpha"):
            w0[k] = np.ones_like(w0[k])
    mb = models.get_model(name, (None, None, 3), nc, losses.ModelLambdaLossType.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER)
    model = mb.model
    assert mb.name.endswith("-MT") and model.input_names[-2:] == ["mt_predictions", "consistency_cost"]
    _compile(model)
    cw = np.ones(nc, dtype=np.float32)
    cw[0] = 0
    images, labels, pw = E.synthetic_batch(B, H, W, nc, seed=seed + 100, block=8, class_weights=cw)
    rng = np.random.RandomState(seed)
    mt_logits = rng.standard_normal((B, H, W, nc)).astype(np.float32) * 2
    masks = E.make_dropout_masks(name, B, seed)
    stream = torch.cuda.Stream()
    cm_total = np.zeros((nc, nc), dtype=np.int64)
    for step, coef in enumerate((2.5, 0.0, 40.0)):
        model.set_weights([w0[k] for k in model.weight_names()])
        model._ensure_buffers(B, H, W)
        g = model.graph
        saved = g.drop_groups
        g.drop_groups = []
        for k, m in masks.items():
            g.drop_views[k].copy_(torch.from_numpy(m.astype(np.float32)))
        ref = E.train_step(name, w0, images, labels, pw, masks, collect=True,
                           mean_teacher=dict(num_unlabeled=1, mt_logits=mt_logits, coefficient=coef))
        model.use_hip_graph = step > 0
        with torch.cuda.stream(stream):
            outs = model.train_on_batch([images, labels, pw, np.ones((B, 1), np.int32), mt_logits, np.full((B, 1), coef, np.float32)],
                                        [np.zeros(B, np.int32), labels[..., None]])
        torch.cuda.synchronize()
        g.drop_groups = saved
        _close(outs[0], ref["loss"], "loss (coef {})".format(coef), atol_rel=1e-6)
        _close(g.dlogits, ref["dlogits"], "dlogits (coef {})".format(coef))
        cost = K.mean_teacher_consistency_cost(torch.tensor(ref["logits"]), torch.tensor(mt_logits, dtype=torch.float64), coef)
        _close(model.consistency_cost(), float(cost), "consistency term", atol_rel=1e-6)
        for k, gr in ref["grads"].items():
            got = g.store.gviews[k].cpu().numpy().astype(np.float64)
            scale = np.abs(gr).max()
            assert np.abs(got - gr).max() <= 1e-3 * scale + 2e-6, "grad {} (coef {})".format(k, coef)
        lab0 = labels.copy()
        lab0[B - 1:] = 0
        cm_total += K.confusion_matrix_np(lab0, g.logits.cpu().numpy(), nc, [0])
        assert np.array_equal(outs[6], cm_total)


def test_trainer_teacher_follows_the_student():
    """SegmentationTrainer in mean-teacher mode: the teacher's logits enter the student's loss, and after every step the
    teacher equals the numpy EMA of (previous teacher, updated student) bit for bit; validation batches carry a zero
    consistency term."""
    trainers = pkg("trainers")
    cfg = {"input_shape": [None, None, 3], "num_labeled_per_batch": 2, "num_unlabeled_per_batch": 1, "random_seed": 5,
           "optimizer": {"name": "adam", "learning_rate": 1e-3},
           "mean_teacher_params": {"ema_smoothing_coefficient_function": "lambda x: 0.5 if x < 1 else 0.9",
                                   "consistency_cost_coefficient_function": "lambda x: 10.0 * min(np.exp(-5.0 * (1.0 - float(x + 1) / 2.0) ** 2), 1.0)"}}
    tr = trainers.SegmentationTrainer("segmentation_semi_supervised_mean_teacher", "enet-naive-upsampling", "t", config=cfg,
                                      num_classes=24)
    assert tr.using_mean_teacher_method and tr.teacher_model is not None
    student, teacher = tr.model, tr.teacher_model
    assert all(np.array_equal(a, b) for a, b in zip(student.get_weights(), teacher.get_weights()))
    B, H, W = 3, 32, 32
    for step in range(2):
        images, labels, pw = E.synthetic_batch(B, H, W, 24, seed=50 + step, block=8)
        t_images = images + np.random.RandomState(step).standard_normal(images.shape).astype(np.float32) * 0.05
        x = [images, labels, pw, np.ones((B, 1), np.int32), t_images]
        t_before = teacher.get_weights()
        x2, _ = tr.modify_batch_data(step, x, None, validation=False)
        assert len(x2) == 6 and tuple(x2[4].shape) == (B, H, W, 24)
        coef = float(tr.consistency_cost_coefficient_function(step))
        assert np.allclose(x2[5], coef)
        mt_host = x2[4].cpu().numpy()
        outs = student.train_on_batch(x2, None)
        # the consistency term of the step is the oracle's value for the logits the kernels saw
        ref_cost = K.mean_teacher_consistency_cost(torch.tensor(student.graph.logits.cpu().numpy(), dtype=torch.float64),
                                                   torch.tensor(mt_host, dtype=torch.float64), coef)
        _close(student.consistency_cost(), float(ref_cost), "consistency term", atol_rel=1e-6)
        assert np.isfinite(outs[0])
        tr.on_batch_end(step)
        a = 0.5 if step < 1 else 0.9
        expect = K.ema_update_np(t_before, student.get_weights(), a)
        got = teacher.get_weights()
        assert all(np.array_equal(e, g_) for e, g_ in zip(expect, got)), "teacher != EMA at step {}".format(step)
    xv, _ = tr.modify_batch_data(2, [images, labels, pw, np.zeros((B, 1), np.int32)], None, validation=True)
    outs = student.test_on_batch(xv, None)
    assert student.consistency_cost() == 0.0 and np.isfinite(outs[0])
    ev = tr.evaluate_teacher(iter([([images, labels, pw, np.zeros((B, 1), np.int32)], None)]), 1)
    assert ev["images"] == B and np.isfinite(ev["loss"])


def test_superpixel_lambda_loss_golden_vector():
    losses = pkg("losses")
    g = np.load(os.path.join(GOLD, "ops.npz"))
    args = [_t(g["sp_logits"]), _t(g["sp_labels"], torch.int32), _t(g["sp_weights"]),
            torch.full((4, 1), 2, dtype=torch.int32, device=DEV), torch.full((4, 1), 1.75, device=DEV)]
    loss = losses.segmentation_superpixel_lambda_loss(args)
    _close(loss, g["sp_loss"], "superpixel loss")
    _close(losses.last_gradient(), g["sp_dlogits"], "superpixel dlogits")
    with pytest.raises(ValueError):
        losses.segmentation_superpixel_lambda_loss(args[:4])


@pytest.mark.parametrize("C,shape,nu,with_mt", [(24, (3, 17, 13), 2, False), (24, (4, 8, 24), 1, True), (5, (2, 9, 7), 1, True),
                                                (24, (2, 6, 6), 0, False), (3, (3, 1, 5), 2, False)])
def test_superpixel_kernels_against_oracle(C, shape, nu, with_mt):
    """Cost and gradient incl. image borders (zero padding per image), the one-sided clip, 1-pixel-high images, batches
    without unlabeled rows, and accumulation on top of the mean-teacher gradient."""
    losses = pkg("losses")
    rng = np.random.RandomState(C + nu)
    B = shape[0]
    logits = rng.standard_normal(shape + (C,)) * 3
    teacher = rng.standard_normal(shape + (C,)) * 3
    labels = rng.randint(0, C, size=shape)
    if nu:
        labels[B - nu:] = rng.randint(0, 3, size=(nu,) + shape[1:])
    w = rng.uniform(0.2, 2.0, size=shape)
    lg = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    if with_mt:
        ref = K.segmentation_mean_teacher_superpixel_lambda_loss(lg, torch.tensor(labels), torch.tensor(w), nu, torch.tensor(teacher), 0.8, 2.5)
    else:
        ref = K.segmentation_superpixel_lambda_loss(lg, torch.tensor(labels), torch.tensor(w), nu, 2.5)
    (dref,) = torch.autograd.grad(ref, [lg])
    args = [_t(logits), _t(labels, torch.int32), _t(w), torch.full((B, 1), nu, dtype=torch.int32, device=DEV)]
    if with_mt:
        args += [_t(teacher), torch.full((B, 1), 0.8, device=DEV)]
    args += [torch.full((B, 1), 2.5, device=DEV)]
    fn = losses.segmentation_mean_teacher_superpixel_lambda_loss if with_mt else losses.segmentation_superpixel_lambda_loss
    loss = fn(args)
    _close(loss, ref.item(), "loss")
    _close(losses.last_gradient(), dref.numpy(), "dlogits")


@pytest.mark.parametrize("with_mt", [False, True])
def test_superpixel_student_step_against_oracle(with_mt):
    """A whole step of an ENet student with the superpixel (and mean-teacher) lambda loss, two unlabeled rows."""
    models, losses = pkg("models"), pkg("losses")
    name, nc, B, H, W, seed = "enet-naive-upsampling", 24, 3, 32, 32, 13
    w0 = E.init_weights(name, nc, seed=seed, perturb=True)
    for k in w0:
        if k.endswith("/alpha"):
            w0[k] = np.ones_like(w0[k])
    T = losses.ModelLambdaLossType
    mb = models.get_model(name, (None, None, 3), nc, T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL if with_mt
                          else T.SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL)
    model = mb.model
    assert mb.name.endswith("-MT-SP" if with_mt else "-SP") and model.input_names[-1] == "unlabeled_cost_coeff"
    _compile(model)
    cw = np.ones(nc, dtype=np.float32)
    cw[0] = 0
    images, labels, pw = E.synthetic_batch(B, H, W, nc, seed=seed + 100, block=8, class_weights=cw)
    rng = np.random.RandomState(seed)
    labels[1:] = np.repeat(np.repeat(rng.randint(0, 3, size=(2, H // 4, W // 4)), 4, axis=1), 4, axis=2)
    mt_logits = rng.standard_normal((B, H, W, nc)).astype(np.float32) * 2
    masks = E.make_dropout_masks(name, B, seed)
    stream = torch.cuda.Stream()
    cm_total = np.zeros((nc, nc), dtype=np.int64)
    for step, (mtc, spc) in enumerate(((1.5, 4.0), (0.0, 30.0))):
        model.set_weights([w0[k] for k in model.weight_names()])
        model._ensure_buffers(B, H, W)
        g = model.graph
        saved = g.drop_groups
        g.drop_groups = []
        for k, m in masks.items():
            g.drop_views[k].copy_(torch.from_numpy(m.astype(np.float32)))
        ref = E.train_step(name, w0, images, labels, pw, masks, collect=True,
                           mean_teacher=dict(num_unlabeled=2, mt_logits=mt_logits, coefficient=mtc) if with_mt else None,
                           superpixel=dict(num_unlabeled=2, coefficient=spc))
        model.use_hip_graph = step > 0
        x = [images, labels, pw, np.full((B, 1), 2, np.int32)]
        if with_mt:
            x += [mt_logits, np.full((B, 1), mtc, np.float32)]
        x += [np.full((B, 1), spc, np.float32)]
        with torch.cuda.stream(stream):
            outs = model.train_on_batch(x, None)
        torch.cuda.synchronize()
        g.drop_groups = saved
        _close(outs[0], ref["loss"], "loss (step {})".format(step), atol_rel=1e-6)
        _close(g.dlogits, ref["dlogits"], "dlogits (step {})".format(step))
        cost = K.unlabeled_superpixel_cost(torch.tensor(labels[1:].astype(np.int64)), torch.tensor(ref["logits"][1:]), spc)
        _close(model.superpixel_cost(), float(cost), "superpixel term", atol_rel=1e-6)
        for k, gr in ref["grads"].items():
            got = g.store.gviews[k].cpu().numpy().astype(np.float64)
            scale = np.abs(gr).max()
            assert np.abs(got - gr).max() <= 1e-3 * scale + 2e-6, "grad {} (step {})".format(k, step)
        lab0 = labels.copy()
        lab0[1:] = 0
        cm_total += K.confusion_matrix_np(lab0, g.logits.cpu().numpy(), nc, [0])
        assert np.array_equal(outs[6], cm_total)


def test_trainer_superpixel_inputs():
    trainers = pkg("trainers")
    cfg = {"input_shape": [None, None, 3], "random_seed": 5, "optimizer": {"name": "adam", "learning_rate": 1e-3},
           "mean_teacher_params": {"ema_smoothing_coefficient_function": "lambda x: 0.99", "consistency_cost_coefficient_function": "lambda x: 2.0"},
           "superpixel_params": {"unlabeled_cost_coefficient_function": "lambda x: 0.5 * (x + 1)"}}
    tr = trainers.SegmentationTrainer("segmentation_semi_supervised_mean_teacher_superpixel", "enet-naive-upsampling", "t",
                                      config=cfg, num_classes=24)
    assert tr.using_mean_teacher_method and tr.using_superpixel_method
    B, H, W = 2, 32, 32
    images, labels, pw = E.synthetic_batch(B, H, W, 24, seed=3, block=8)
    labels[1] = (labels[1] % 3)
    x, _ = tr.modify_batch_data(1, [images, labels, pw, np.ones((B, 1), np.int32), images], None, validation=False)
    assert len(x) == 7 and np.allclose(x[5], 2.0) and np.allclose(x[6], 1.0)
    outs = tr.model.train_on_batch(x, None)
    tr.on_batch_end(1)
    assert np.isfinite(outs[0]) and tr.model.superpixel_cost() > 0 and tr.model.consistency_cost() >= 0
    xv, _ = tr.modify_batch_data(2, [images, labels, pw, np.zeros((B, 1), np.int32)], None, validation=True)
    assert len(xv) == 7 and not np.any(xv[5]) and not np.any(xv[6])
    tr.model.test_on_batch(xv, None)
    assert tr.model.superpixel_cost() == 0.0 and tr.model.consistency_cost() == 0.0
