"""GPU parity: full training steps (forward, loss, backward, BN moving updates, Adam) of the HIP path against the CPU
oracle on identical weights, inputs and dropout masks.

Tolerances: rtol 1e-3 for fp32 activations, loss and gradients, measured per tensor against
the tensor's own scale; pooling indices and confusion matrices bit-exact.  Forward quantities are checked against the
fp64 oracle at that tolerance.

Gradients: the network is piecewise linear (PReLU kinks at 0, max-pool selections), so two correct evaluations can take
different branches at the handful of pre-activations that sit within rounding distance of a kink, and on these tiny
shapes (BN over 128 samples) one flipped element moves every upstream gradient by up to a few percent (the fp32 oracle
against the fp64 oracle shows the same jumps).  Every case therefore replays the branches the kernels took: the sign of
every PReLU / LeakyReLU input is recomputed on the host from the device's own raw tensors and statistics (same float32 operation order
as the kernels, no fused multiply-add, so the signs are the kernels' own) and the pooling selections are read back, and
the fp64 oracle is evaluated with exactly those branches (oracle.enet_oracle.train_step(forced=...)).  Against that
evaluation every gradient tensor must hold rtol 1e-3 of its scale - with random PReLU slopes as well as in the `smooth`
cases (every alpha 1, no kink), for the ENets and for U-Net / SegNet alike.
Adam: the update the device applied (w_new - w_old) must equal Adam's formula evaluated in float64 on the DEVICE's own
gradient and the oracle's slot values (rtol 1e-3): that checks arena wiring, step counter and scaling through a real
step, while the gradients themselves are checked above.
Per-operator gradient parity at rtol 1e-3 with random alphas is covered by tests/test_ops_gpu.py."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _keep_expansion_outputs(monkeypatch):
    """The branch replay below reads every bottleneck's expansion output y3 from the device; the training step does not store
    it for regular bottlenecks unless asked to (the forward kernel then writes it beside the block output, same values)."""
    monkeypatch.setenv("SSMI_KEEP_Y3", "1")



def _close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), "{}: {} of {} elements off; max abs err {:.3e} (ref scale {:.3e})".format(
        what, int(bad.sum()), a.size, float(err.max()), float(np.abs(b).max()))


def _f32(t):
    return t.detach().float().cpu().numpy()


def _hip_branches(model, w_before):
    """Signs of every PReLU input and the pooling selections of the step the device just ran (ENet graphs).  `w_before`:
    the weights the step started from (the optimizer has already moved the arena)."""
    eng = pkg("engine")
    out = {}

    def _bn_z(y, bn):
        # gamma * ((y - mean) * invstd) + beta in float32, the operation order of the kernels
        g, b = np.asarray(w_before[bn.name + "/gamma"], np.float32), np.asarray(w_before[bn.name + "/beta"], np.float32)
        return g * ((_f32(y) - _f32(bn.mean)) * _f32(bn.invstd)) + b

    for b in model.graph.blocks:
        if isinstance(b, eng.InitialBlock):
            if b.bn is not None:
                out["initial_block_prelu"] = np.sign(_bn_z(b.cat, b.bn))
        elif isinstance(b, eng.Bottleneck):
            p = b.name
            out[p + "_prelu_1"] = np.sign(_bn_z(b.y1, b.bn1))
            out[p + "_prelu_2"] = np.sign(_bn_z(b.y2, b.bn2))
            s = _bn_z(b.y3, b.bn3)
            if b._drop_used is not None:
                s = s * _f32(b._drop_used)[:, None, None, :]
            if b.mode == 0:
                s = s + _f32(b.x)
            elif b.mode == 1:
                s[..., :b.cin] = s[..., :b.cin] + _f32(b.pooled)
                out[p + "_other_pool2d/argmax"] = b.idx.cpu().numpy().astype(np.int64)
            else:
                o = _bn_z(b.yo, b.bn_o)
                if b.mode == 2:
                    o = o.repeat(2, axis=1).repeat(2, axis=2)
                else:
                    o = E.K.max_unpool_np(o, b.unpool_idx.cpu().numpy().astype(np.int64))
                s = s + o
            out[p + "_prelu_3"] = np.sign(s)
        elif isinstance(b, eng.EnhancedHead):
            out["final_nn_rs_conv_prelu"] = np.sign(_bn_z(b.ya, b.bn))
        elif hasattr(b, "units"):      # U-Net / SegNet: LeakyReLU signs of every activated unit, max-pool selections
            for u in b.units:
                if u.act != eng.ACT_NONE:
                    out[u.name + "_activation"] = np.sign(_bn_z(u.y, u.bn))
            if hasattr(b, "pooled"):
                out[b.name + "_pool/argmax"] = b.idx.cpu().numpy().astype(np.int64)
    return out


def _run_case(model_name, B, H, W, seed, steps=2, smooth=False, monkeypatch=None):
    models = pkg("models")
    losses = pkg("losses")
    metrics = pkg("metrics")
    opt = pkg("keras_extensions.extended_optimizers")
    nc = 24
    w0 = E.init_weights(model_name, nc, seed=seed, perturb=True)
    if smooth:
        for k in w0:
            if k.endswith("/alpha"):
                w0[k] = np.ones_like(w0[k])
        if model_name in E.VGG_STYLE:
            # LeakyReLU slope 1 on both sides (the slope is a builder constant, src/models.py:840)
            monkeypatch.setattr(E.V, "LEAKY_ALPHA", 1.0)
            monkeypatch.setattr(pkg("engine_unet"), "LEAKY_ALPHA", 1.0)
    mb = models.get_model(model_name, (None, None, 3), nc, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY)
    model = mb.model
    model.keep_logits = True      # the ENet steps compute the logits inside the loss kernel; this test compares them
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
        ref64 = E.train_step(model_name, weights, images, labels, pw, masks, collect=True)
        prev_state = {"iterations": state["iterations"], "m": dict(state["m"]), "v": dict(state["v"])}
        ref = E.train_step(model_name, weights, images, labels, pw, masks, adam=adam, adam_state=state, collect=True,
                           dtype=torch.float32)
        with torch.cuda.stream(stream):
            # fetch pre-update quantities with an eager run of the same segments
            model.use_hip_graph = (step > 0)
            outs = model.train_on_batch([images, labels, pw, np.zeros((B, 1), dtype=np.int32)],
                                        [np.zeros(B, dtype=np.int32), labels[..., None]])
        torch.cuda.synchronize()
        g.drop_groups = saved_groups
        hip_logits = g.logits.cpu().numpy()
        hip_grads = {k: g.store.gviews[k].cpu().numpy().astype(np.float64) for k in ref64["grads"]}
        ref64g = ref64
        if True:
            # the fp64 oracle along the branches the kernels took: the reference for the gradients (forward quantities stay
            # with the oracle's own branches: a flipped element moves them by (1 - alpha) * |z| only)
            ref64g = E.train_step(model_name, weights, images, labels, pw, masks, forced=_hip_branches(model, weights))
        # the matrix must be bit-exact for the logits the kernel actually saw ...
        cm_total += E.K.confusion_matrix_np(labels, hip_logits, nc, [0])
        # ... and may differ from the oracle's own matrix only at pixels whose two best fp64 logits are a near-tie
        srt = np.sort(ref64["logits"].reshape(-1, nc), axis=-1)
        near_ties = int(((srt[:, -1] - srt[:, -2]) < 1e-4 * np.abs(ref64["logits"]).max()).sum())
        moved = int(np.abs(E.K.confusion_matrix_np(labels, hip_logits, nc, [0]) - ref64["cm"]).sum()) // 2
        assert moved <= near_ties, "{} pixels classified differently from the fp64 oracle, {} near-ties".format(moved, near_ties)
        # loss / logits / dlogits
        _close(outs[0], ref64["loss"], 1e-3, 1e-6, "loss step {}".format(step))
        _close(g.logits.cpu().numpy(), ref64["logits"], 1e-3, 1e-4 * np.abs(ref64["logits"]).max(), "logits")
        _close(g.dlogits.cpu().numpy(), ref64["dlogits"], 1e-3, 1e-4 * np.abs(ref64["dlogits"]).max(), "dlogits")

        # pooling indices bit-exact
        for bname in ("en_bn_1.0", "en_bn_2.0"):
            key = bname + "_other_pool2d/argmax"
            if key in ref["acts"]:
                got = g.by_name[bname].idx.cpu().numpy().astype(np.int64)
                assert np.array_equal(got, ref["acts"][key]), "argmax indices of {} differ".format(bname)
        # gradients: per tensor, tolerance relative to the tensor's own scale
        gv = g.store.gviews
        n_tight = 0
        worst = (0.0, "", 0.0, 0.0, 0.0)
        for k, gr in ref64g["grads"].items():
            got = gv[k].cpu().numpy().astype(np.float64)
            scale = np.abs(gr).max()
            e_hip = np.abs(got - gr).max()
            e_f32 = np.abs(ref["grads"][k].astype(np.float64) - gr).max()
            # 2e-6: biases in front of a BN have an exactly-zero gradient, both sides return rounding noise
            tol = 1e-3 * scale + 2e-6
            assert e_hip <= tol, "grad {}: err {:.3e} > tol {:.3e} (scale {:.3e}, fp32-oracle err {:.3e})".format(
                k, e_hip, tol, scale, e_f32)
            n_tight += e_hip <= 1e-3 * scale + 2e-6
            if e_hip / tol > worst[0]:
                worst = (e_hip / tol, k, e_hip, scale, e_f32)
        print("{} step {}: {} of {} gradient tensors within rtol 1e-3 of the fp64 oracle; closest to its bound: {} "
              "(err {:.3e} = {:.0%} of the bound, scale {:.3e}, fp32-oracle err {:.3e})".format(
                  model_name, step, n_tight, len(ref64["grads"]), worst[1], worst[2], worst[0], worst[3], worst[4]))
        # confusion matrix bit-exact (cumulative)
        same_cm = bool(np.array_equal(outs[6], cm_total))
        assert same_cm, "confusion matrix differs from the one recomputed from the kernel's own logits"
        m = metrics.metrics_from_confusion_matrix(cm_total, [0])
        assert abs(outs[3] - m["accuracy"]) < 1e-12
        # updated weights (Adam + BN moving statistics)
        new_w = model.get_weights_dict()
        tkeys = list(ref64["grads"].keys())
        ps = [torch.tensor(np.asarray(weights[k], dtype=np.float32).astype(np.float64)) for k in tkeys]
        gs = [torch.tensor(hip_grads[k]) for k in tkeys]
        ms = [prev_state["m"][k].double().clone() if k in prev_state["m"] else torch.zeros_like(p) for k, p in zip(tkeys, ps)]
        vs = [prev_state["v"][k].double().clone() if k in prev_state["v"] else torch.zeros_like(p) for k, p in zip(tkeys, ps)]
        expect = E.K.adam_step([p.clone() for p in ps], gs, ms, vs, prev_state["iterations"], **adam)
        for k, p_old, p_new in zip(tkeys, ps, expect):
            d_ref = (p_new - p_old).numpy()
            d_hip = new_w[k].astype(np.float64) - p_old.numpy()
            # 1e-3 of the update + the float32 rounding of the stored weight (the update is a difference of two of them)
            tol = 1e-3 * np.abs(d_ref) + 2.0 ** -22 * np.maximum(np.abs(p_old.numpy()), 1e-3)
            bad = np.abs(d_hip - d_ref) > tol
            assert not bad.any(), "Adam update of {}: {} of {} elements off, worst {:.3e} vs {:.3e}".format(
                k, int(bad.sum()), bad.size, float(d_hip.reshape(-1)[np.argmax(np.abs(d_hip - d_ref))]),
                float(d_ref.reshape(-1)[np.argmax(np.abs(d_hip - d_ref))]))
        for k, v in ref["new_weights"].items():
            if k.endswith(("moving_mean", "moving_variance")):
                _close(new_w[k], v, 1e-3, 1e-4 * np.abs(v).max() + 1e-6, "updated " + k)
        # continue from the ORACLE's weights and optimizer slots so that errors do not compound across steps
        weights = ref["new_weights"]
        model.set_weights([weights[k] for k in model.weight_names()])
        specs = g.store.specs
        for slot, arena in (("m", model.optimizer.m), ("v", model.optimizer.v)):
            for k in tkeys:
                sp = specs[k]
                arena[sp.offset:sp.offset + sp.size].copy_(state[slot][k].reshape(-1).to(torch.float32))


def test_enet_naive_train_step_parity_smooth():
    _run_case("enet-naive-upsampling", 2, 64, 64, seed=3, steps=2, smooth=True)


def test_enet_max_unpooling_train_step_parity_smooth():
    _run_case("enet-max-unpooling", 2, 64, 96, seed=5, steps=2, smooth=True)


def test_enet_enhanced_train_step_parity_smooth():
    _run_case("enet-naive-upsampling-enhanced", 2, 64, 64, seed=7, steps=2, smooth=True)


def test_enet_enhanced_train_step_parity():
    _run_case("enet-naive-upsampling-enhanced", 2, 48, 80, seed=9, steps=1)


def test_enet_naive_train_step_parity():
    _run_case("enet-naive-upsampling", 2, 64, 64, seed=3, steps=2)


def test_enet_naive_train_step_parity_ragged_tiles():
    """Pixel counts that end in a partial 32- / 64-pixel tile of the expansion kernels at every stage (180 and 720 pixels), so
    the chained / carried-through passes run their masked last tiles inside a whole step."""
    _run_case("enet-naive-upsampling", 3, 48, 80, seed=11, steps=1)


@pytest.mark.parametrize("model", ["enet-naive-upsampling", "enet-max-unpooling"])
def test_enet_train_step_parity_odd_stage_sizes(model):
    """40 x 72 is a multiple of 8 and not of 16: stages 2 / 3 work on 5 x 9 pixels per image (odd in both directions, 45 pixels),
    the smallest granularity three 2x reductions allow."""
    _run_case(model, 2, 40, 72, seed=13, steps=1)


def test_enet_max_unpooling_train_step_parity():
    _run_case("enet-max-unpooling", 3, 64, 96, seed=5, steps=1)


def test_unet_train_step_parity_smooth(monkeypatch):
    _run_case("unet", 2, 32, 32, seed=11, steps=1, smooth=True, monkeypatch=monkeypatch)


def test_segnet_train_step_parity_smooth(monkeypatch):
    _run_case("segnet", 2, 32, 48, seed=13, steps=2, smooth=True, monkeypatch=monkeypatch)


def test_segnet_train_step_parity():
    _run_case("segnet", 3, 32, 32, seed=15, steps=1)


def test_unet_train_step_parity():
    _run_case("unet", 2, 48, 32, seed=17, steps=1)


def test_frozen_layers_and_optimizer_state_roundtrip(tmp_path):
    """Layers marked untrainable before compile() take no updates; Model.save + load_optimizer_weights restore the Adam
    slots and the iteration counter so that a resumed step equals the uninterrupted one."""
    models, losses, opt = pkg("models"), pkg("losses"), pkg("keras_extensions.extended_optimizers")
    name, nc = "enet-naive-upsampling", 24
    w0 = E.init_weights(name, nc, seed=4, perturb=True, dtype=np.float32)
    images, labels, pw = E.synthetic_batch(2, 32, 32, nc, seed=5, block=8)
    x = [images, labels, pw, np.zeros((2, 1), dtype=np.int32)]
    y = [np.zeros(2, dtype=np.int32), labels[..., None]]

    def make(frozen_layers=()):
        mb = models.get_model(name, (None, None, 3), nc, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY)
        m = mb.model
        m.set_weights([w0[k] for k in m.weight_names()])
        for i in frozen_layers:
            m.layers[i].trainable = False
        m.compile(optimizer=opt.Adam(lr=1e-3), loss={'loss': losses.dummy_loss, 'logits': lambda t, p: 0 * p},
                  loss_weights={'loss': 1.0, 'logits': 0.0})
        m.use_hip_graph = False
        return mb, m

    mb, m = make(frozen_layers=range(0, 6))
    with torch.cuda.stream(torch.cuda.Stream()):
        m.train_on_batch(x, y)
    torch.cuda.synchronize()
    after = m.get_weights_dict()
    store = m.graph.store
    frozen_keys = {k for l in m.layers[:6] for k in l.weight_keys}
    moved = 0
    for k in store.trainable_keys():
        if k in frozen_keys:
            assert np.array_equal(after[k], w0[k]), "frozen weight {} changed".format(k)
        else:
            moved += int(not np.array_equal(after[k], w0[k]))
    assert moved > 300

    # resume: A runs two steps; B runs one, is saved, a fresh model restores weights + optimizer state and runs the second
    _, a = make()
    _, b = make()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for mdl in (a, b):
            mdl.seed = 99
        a.train_on_batch(x, y)
        a.train_on_batch(x, y)
        b.train_on_batch(x, y)
        torch.cuda.synchronize()
        path = str(tmp_path / "ckpt.npz")
        b.save(path)
        mbc, c = make()
        c.seed = 99
        c.load_weights(path)
        assert mbc.load_optimizer_weights(path) is True
        assert c.optimizer.iterations == 1
        c.train_on_batch(x, y)
    torch.cuda.synchronize()
    wa, wc = a.get_weights_dict(), c.get_weights_dict()
    for k in wa:
        assert np.array_equal(wa[k], wc[k]), "resumed step differs in {}".format(k)
    assert mbc.load_optimizer_weights(str(tmp_path / "ckpt.npz")) is True
    c.save_weights(str(tmp_path / "weights_only.npz"))
    assert mbc.load_optimizer_weights(str(tmp_path / "weights_only.npz")) is False


def test_trainer_checkpoint_resume(tmp_path):
    """save_model_weights / _load_latest_weights_for_model (src/trainers.py:1209-1238, 1367-1374): the newest checkpoint is
    found by its epoch, restored bit for bit, and training continues from the following epoch."""
    trainers = pkg("trainers")
    cfg = {"input_shape": [None, None, 3], "random_seed": 8, "optimizer": {"name": "sgd", "learning_rate": 0.01, "momentum": 0.9},
           "keras_model_checkpoint": {"checkpoint_file_path": str(tmp_path) + "/{model_folder}/checkpoints/weights.{epoch:02d}-{val_loss:.2f}.hdf5"},
           "optimizer_checkpoint_file_path": str(tmp_path) + "/{model_folder}/checkpoints/optimizer/optimizer-checkpoint.json"}
    tr = trainers.SegmentationTrainer("segmentation_supervised", "enet-max-unpooling", "run1", config=cfg, num_classes=24)
    images, labels, pw = E.synthetic_batch(2, 32, 32, 24, seed=1, block=8)
    x = [images, labels, pw, np.zeros((2, 1), np.int32)]
    tr.model.train_on_batch(x, None)
    p1 = tr.save_model_weights(epoch_index=1, val_loss=0.5)
    w1 = tr.model.get_weights()
    tr.model.train_on_batch(x, None)
    p4 = tr.save_model_weights(epoch_index=4, val_loss=0.25)
    w4 = tr.model.get_weights()
    assert p1.endswith("run1/checkpoints/weights.01-0.50.hdf5") and p4.endswith("weights.04-0.25.hdf5")
    oc = tr.save_optimizer_settings(tr.model)
    import json
    assert json.load(open(oc))["momentum"] == 0.9
    tr2 = trainers.SegmentationTrainer("segmentation_supervised", "enet-max-unpooling", "run1", config=cfg, num_classes=24)
    assert tr2._load_latest_weights_for_model(tr2.model, p1) == 5
    assert all(np.array_equal(a, b) for a, b in zip(tr2.model.get_weights(), w4))
    assert not all(np.array_equal(a, b) for a, b in zip(w1, w4))
    assert tr2._load_latest_weights_for_model(tr2.model, str(tmp_path / "nothing" / "here.hdf5")) == 0
