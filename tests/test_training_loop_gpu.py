"""The training loop around the step: optimizer hyper-parameters that change between replays of a captured step, fresh
teacher noise per prediction, the rows the training metrics cover, checkpoint + bit-exact resume through the trainer's
callbacks, and a 30-step loss trajectory against the fp32 CPU oracle.

Tolerances: graph replay against eager launches and resume against an uninterrupted run are bit-exact (same kernels, same
order); the loss trajectory holds rtol 1e-3 for the first steps (fp32 kernels against the fp32 oracle, both from identical
weights) and must fall over the 30 steps."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E
from oracle import tf_keras_ops as K

pytestmark = pytest.mark.gpu
NC = 24


def _model(name="enet-naive-upsampling", seed=3, lr=1e-3, opt="adam", metrics_nu=0, ignore=(0,)):
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    w0 = E.init_weights(name, NC, seed=seed, perturb=True)
    m = models.get_model(name, (None, None, 3), NC, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY).model
    m.set_weights([w0[k] for k in m.weight_names()])
    o = xo.Adam(lr=lr) if opt == "adam" else xo.SGD(lr=lr, momentum=0.9)
    ic = list(ignore)
    m.compile(optimizer=o, metrics={'logits': [metrics.segmentation_accuracy(metrics_nu, ic),
                                               metrics.segmentation_confusion_matrix(NC, metrics_nu, ic)]})
    return m, w0


@pytest.mark.parametrize("opt", ["adam", "sgd"])
def test_hyperparameter_change_reaches_the_replayed_graph(opt):
    """lr (and beta_2 / momentum) set on the optimizer after the step was captured must act on the next replay: the
    graph-replayed model follows an eager twin bit for bit, and differs from a twin whose rate was left alone."""
    B, H, W = 2, 32, 32
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=5, block=8)
    x = [images, labels, pw, np.zeros((B, 1), np.int32)]
    runs = {}
    for tag, use_graph, change in (("graph", True, True), ("eager", False, True), ("unchanged", True, False)):
        m, _ = _model(opt=opt)
        m.use_hip_graph = use_graph
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m.train_on_batch(x)
            m.train_on_batch(x)           # the second call replays the captured step
            w_before = m.graph.store.theta.clone()
            if change:
                m.optimizer.lr = m.optimizer.lr * 0.5
                if opt == "adam":
                    m.optimizer.beta_2 = 0.99
                else:
                    m.optimizer.momentum = 0.5
            m.train_on_batch(x)
            torch.cuda.synchronize()
        runs[tag] = (w_before.cpu().numpy(), m.graph.store.theta.cpu().numpy())
        m.close()
    assert np.array_equal(runs["graph"][0], runs["eager"][0]) and np.array_equal(runs["graph"][0], runs["unchanged"][0])
    assert np.array_equal(runs["graph"][1], runs["eager"][1]), "replayed step ignored the new hyper-parameters"
    d_new = runs["graph"][1] - runs["graph"][0]
    d_old = runs["unchanged"][1] - runs["unchanged"][0]
    assert np.abs(d_new).max() > 0 and not np.array_equal(d_new, d_old)


def test_training_phase_predictions_draw_fresh_noise():
    """predict_on_batch(use_training_phase_layers=True) of a model that is never stepped (the mean teacher): two
    predictions of one batch differ and so do the dropout tables; inference-phase predictions stay identical."""
    m, _ = _model()
    images, _, _ = E.synthetic_batch(2, 32, 32, NC, seed=9, block=8)
    a = m.predict_on_batch(images, use_training_phase_layers=True)
    t1 = m.graph.drop_arena.clone()
    b = m.predict_on_batch(images, use_training_phase_layers=True)
    t2 = m.graph.drop_arena.clone()
    assert not torch.equal(t1, t2) and not np.array_equal(a, b)
    c, d = m.predict_on_batch(images), m.predict_on_batch(images)
    assert np.array_equal(c, d)
    # an uncompiled model too (the reference's teacher before compile)
    models = pkg("models")
    u = models.get_model("enet-naive-upsampling", (None, None, 3), NC).model
    u.graph.store.init_values(1)
    assert not np.array_equal(u.predict_on_batch(images, use_training_phase_layers=True),
                              u.predict_on_batch(images, use_training_phase_layers=True))


def test_training_metrics_cover_labeled_rows_only():
    """Metrics built with num_unlabeled = 1 and no ignored class: in the training phase the confusion matrix counts the
    first B - 1 rows (src/metrics.py:255-259), in the test phase every row."""
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    B, H, W = 3, 32, 32
    w0 = E.init_weights("enet-naive-upsampling", NC, seed=4, perturb=True)
    m = models.get_model("enet-naive-upsampling", (None, None, 3), NC,
                         losses.ModelLambdaLossType.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER).model
    m.set_weights([w0[k] for k in m.weight_names()])
    m.compile(optimizer=xo.Adam(lr=1e-3), metrics={'logits': [metrics.segmentation_confusion_matrix(NC, 1, [])]})
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=6, block=8, class_weights=np.ones(NC, np.float32))
    mt = np.random.RandomState(0).standard_normal((B, H, W, NC)).astype(np.float32)
    x = [images, labels, pw, np.ones((B, 1), np.int32), mt, np.full((B, 1), 2.0, np.float32)]
    outs = m.train_on_batch(x)
    logits = m.graph.logits.cpu().numpy()
    assert np.array_equal(outs[3], K.confusion_matrix_np(labels[:B - 1], logits[:B - 1], NC, []))
    m.reset_metrics()
    outs = m.test_on_batch([images, labels, pw, np.zeros((B, 1), np.int32), np.zeros_like(mt), np.zeros((B, 1), np.float32)])
    logits = m.graph.logits.cpu().numpy()
    assert np.array_equal(outs[3], K.confusion_matrix_np(labels, logits, NC, []))
    assert int(outs[3].sum()) == B * H * W


class _Gen:
    """Deterministic (epoch, batch) -> batch, like the reference's Sequence.get_batch."""

    def __init__(self, B=2, H=32, W=32):
        self.B, self.H, self.W = B, H, W

    def get_batch(self, e, b):
        images, labels, pw = E.synthetic_batch(self.B, self.H, self.W, NC, seed=1000 + 17 * e + b, block=8)
        return [images, labels, pw, np.zeros((self.B, 1), np.int32)], [np.zeros(self.B, np.int32), labels[..., None]]


def _trainer(tmp_path, folder, epochs):
    trainers = pkg("trainers")
    cfg = {"input_shape": [None, None, 3], "num_labeled_per_batch": 2, "random_seed": 5, "num_epochs": epochs,
           "ignore_classes": [0], "optimizer": {"name": "adam", "learning_rate": 1e-3, "decay": 0.0},
           "log_folder_path": str(tmp_path / "logs" / "{model_folder}"),
           "keras_model_checkpoint": {"checkpoint_file_path": str(tmp_path / "{model_folder}" / "weights.{epoch:02d}-{val_loss:.2f}.hdf5"),
                                      "monitor": "val_loss", "save_best_only": False, "save_weights_only": False},
           "keras_csv_log_file_path": "{log_folder_path}/per-epoch-data.csv",
           "optimizer_checkpoint_file_path": str(tmp_path / "{model_folder}" / "optimizer" / "optimizer-checkpoint.json"),
           "stepwise_learning_rate_scheduler": {"lr_schedule": "lambda x: 1e-3 * min(1.0, float(x + 1) / 4.0)",
                                                "b2_schedule": "lambda x: 0.99 if x < 3 else 0.999", "apply_b2_schedule": True},
           "reduce_lr_on_plateau": {"factor": 0.5, "patience": 1}, "early_stopping": {"patience": 50}}
    return trainers.SegmentationTrainer("segmentation_supervised", "enet-naive-upsampling", folder, config=cfg, num_classes=NC)


def test_trainer_checkpoints_every_epoch_and_resumes_bit_exactly(tmp_path):
    gen, steps = _Gen(), 2
    full = _trainer(tmp_path, "full", 3)
    hist = full.train(gen, steps, validation_data=gen, validation_steps=1)
    assert len(hist) == 3 and full.initial_epoch == 0
    files = sorted(os.listdir(tmp_path / "full"))
    assert [f.split("-")[0] for f in files if f.endswith(".hdf5")] == ["weights.00", "weights.01", "weights.02"]
    with open(tmp_path / "full" / "optimizer" / "optimizer-checkpoint.json") as f:
        oc = json.load(f)
    assert oc["beta_2"] == 0.999 and abs(oc["lr"] - 1e-3) < 1e-12      # the schedules reached the optimizer
    csv_rows = open(tmp_path / "logs" / "full" / "per-epoch-data.csv").read().strip().splitlines()
    assert len(csv_rows) == 4 and "val_loss" in csv_rows[0] and "logits_cfm" not in csv_rows[0]
    # the same run in two parts: two epochs, then a new process-equivalent trainer that finds the checkpoints
    part = _trainer(tmp_path, "part", 2)
    part.train(gen, steps, validation_data=gen, validation_steps=1)
    del part
    rest = _trainer(tmp_path, "part", 3)
    rest.train(gen, steps, validation_data=gen, validation_steps=1)
    assert rest.initial_epoch == 2 and rest.model.optimizer.iterations == 3 * steps
    for a, b, k in zip(full.model.get_weights(), rest.model.get_weights(), full.model.weight_names()):
        assert np.array_equal(a, b), "resumed run differs in " + k
    for k, t in full.model.optimizer.slot_tensors().items():
        assert torch.equal(t, rest.model.optimizer.slot_tensors()[k]), "optimizer slot " + k


def test_schedule_strings_are_not_evaluated_as_python():
    S = pkg("schedules")
    f = S.parse_schedule("lambda x: 276.552 * min( np.exp( min(-5.0 * (1.0 - ((float(x)/24000.0)**2)), 1.0) ), 1.0 )")
    assert abs(f(24000) - 276.552) < 1e-9 and abs(f(0) - 276.552 * np.exp(-5.0)) < 1e-9
    for bad in ("lambda x: __import__('os').getcwd()", "lambda x: x.__class__", "lambda x: open('f')", "lambda x: [x][0]", "1 + 1"):
        with pytest.raises(ValueError):
            S.parse_schedule(bad)


def test_thirty_step_loss_trajectory_matches_the_oracle_and_falls():
    """Fixed batch, 30 Adam steps, no dropout noise (all-keep tables).  Adam's first updates are ~lr * sign(g), so rounding
    differences in tiny gradients (elements whose terms cancel almost completely) decide update signs and grow along a
    trajectory: the fp32 CPU oracle itself drifts up to 1e-3 from the fp64 oracle over these steps, and the kernels' long
    fp32 accumulation chains have a higher noise floor than the oracle's blocked sums.  Bound: 1e-3 relative to the fp64
    oracle for the first three steps, 5e-3 for all thirty, the fp32 oracle's own drift reported beside it; the loss falls."""
    name, B, H, W, lr = "enet-naive-upsampling", 2, 64, 64, 2e-3
    m, w0 = _model(name, seed=21, lr=lr)
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=33, block=16)
    x = [images, labels, pw, np.zeros((B, 1), np.int32)]
    m._ensure_buffers(B, H, W)
    m.graph.drop_groups = []            # no refresh: the tables stay at their initial value of one (keep everything)
    masks = {k: np.ones((B, c)) for k, (r, c) in E.dropout_layer_rates(name).items()}
    got, ref = [], {}
    s = torch.cuda.Stream()
    for step in range(30):
        with torch.cuda.stream(s):
            got.append(m.train_on_batch(x)[0])
    for dt in (torch.float32, torch.float64):
        w, state, seq = w0, {"iterations": 0, "m": {}, "v": {}}, []
        for step in range(30):
            r = E.train_step(name, w, images, labels, pw, masks, dtype=dt, adam=dict(lr=lr), adam_state=state)
            seq.append(r["loss"])
            w = r["new_weights"]
        ref[dt] = np.array(seq)
    got, r32, r64 = np.array(got), ref[torch.float32], ref[torch.float64]
    rel = np.abs(got - r64) / np.abs(r64)
    own = np.abs(r32 - r64) / np.abs(r64)
    assert np.all(rel[:3] <= 1e-3), (got[:3], r64[:3])
    assert np.all(rel <= 5e-3), (rel.max(), own.max())
    assert got[-1] < 0.9 * got[0] and np.mean(got[-5:]) < np.mean(got[:5])
    assert r64[-1] < 0.9 * r64[0]


def test_weight_gradient_schedule_does_not_change_results(monkeypatch):
    """Where the weight gradients are forked from the main stream and when their launches are created (engine.RunCfg:
    per-layer fork points, launches issued one block later) is scheduling only: the step with one fork per block and
    immediate launches, the step with everything on the main stream, and the step with one strip reduction per layer instead of
    one grouped launch per join, give the same bits; nothing stays queued."""
    import numpy as np
    import torch
    from conftest import pkg
    from oracle import enet_oracle as E
    eng = pkg("engine")
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    # large enough that kernels of both streams are in flight together (a dependency lost in the schedule would show)
    name, NC, B, H, W = "enet-max-unpooling", 24, 4, 256, 256
    w0 = E.init_weights(name, NC, seed=21, perturb=True)
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=22, block=8)
    res = []
    for lag, per_conv, overlap, grouped in ((1, True, True, False), (0, False, True, False), (2, True, True, False), (1, True, False, False),
                                            (1, True, True, True)):
        monkeypatch.setattr(eng, "_WGRAD_LAG", lag)
        monkeypatch.setattr(eng, "_WGRAD_FORK_PER_CONV", per_conv)
        # grouped: the strip reductions of all layers as one launch per join (ssmi_wg_reduce_defer_*) instead of one per layer
        # (opt-in: measured slower than the per-layer launches)
        monkeypatch.delenv("SSMI_GROUPED_REDUCE", raising=False)
        if grouped:
            monkeypatch.setenv("SSMI_GROUPED_REDUCE", "join")
        m = models.get_model(name, (None, None, 3), NC, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY).model
        m.graph.overlap_weight_gradients = overlap
        m.set_weights([w0[k] for k in m.weight_names()])
        m.compile(optimizer=xo.Adam(lr=1e-3), metrics={'logits': [metrics.segmentation_confusion_matrix(NC, 0, [0])]})
        with torch.cuda.stream(torch.cuda.Stream()):
            outs = [m.train_on_batch([images, labels, pw, np.zeros((B, 1), np.int32)]) for _ in range(3)]
        torch.cuda.synchronize()
        cfg = m.graph._cfg
        assert not cfg._pending and not cfg._deferred, "weight gradients left unlaunched"
        assert (cfg.side is not None) == overlap
        assert cfg.defer_reduce == (overlap and grouped) and not cfg._reduce_open
        res.append((outs, [w.copy() for w in m.get_weights()]))
    for outs, ws in res[1:]:
        for a, b in zip(res[0][0], outs):
            for u, v in zip(a, b):
                assert np.array_equal(np.asarray(u), np.asarray(v))
        for u, v in zip(res[0][1], ws):
            assert np.array_equal(u, v)
