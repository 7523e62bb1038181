"""Host logic of the training loop (no GPU): epoch logs as the reference's ExtendedBaseLogger reports them
(src/keras_extensions/extended_callbacks.py:35-63), what the callbacks see, the data-parallel log reduction, the
StepwiseLearningRateScheduler's beta_2 default and the schedule interpreter's Python-2 arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg, ROOT


class _Gen:
    def __init__(self, sizes):
        self.sizes = sizes

    def get_batch(self, e, b):
        n = self.sizes[b % len(self.sizes)]
        return [np.zeros((n, 8, 8, 3), np.float32)], [None]


def _stub_model(losses, val_losses=None, cms=None):
    """An ExtendedModel whose device step is replaced by canned per-batch outputs [loss, loss, 0, streaming metric]."""
    xm = pkg("keras_extensions.extended_model")
    m = object.__new__(xm.ExtendedModel)
    m.metrics_names = ["loss", "loss_loss", "logits_loss", "logits_acc"]
    m.metrics_streaming = ["logits_acc"]
    m.metrics_excluded_from_callbacks = []
    m._metric_fns = []
    m._cm = None
    m.log_sync = None
    m.stop_training = False
    m._graphs = {}
    state = {"t": 0, "v": 0}

    def train_on_batch(x, y=None):
        v = losses[state["t"] % len(losses)]
        state["t"] += 1
        return [v, v, 0.0, 0.1 * state["t"]]

    def test_on_batch(x, y=None):
        v = (val_losses or losses)[state["v"] % len(val_losses or losses)]
        state["v"] += 1
        return [v, v, 0.0, 0.01 * state["v"]]
    m.train_on_batch, m.test_on_batch = train_on_batch, test_on_batch
    m.reset_metrics = lambda: None
    return m


def test_epoch_logs_are_sample_weighted_means_not_the_last_batch():
    cb = pkg("callbacks")
    seen = []

    class Spy(cb.Callback):
        def on_epoch_end(self, epoch, logs=None):
            seen.append(dict(logs))

    m = _stub_model([1.0, 3.0], val_losses=[10.0, 20.0, 60.0])
    hist = m.fit_generator(_Gen([2, 6]), 2, epochs=1, callbacks=[Spy()], validation_data=_Gen([1, 1, 2]), validation_steps=3)
    # batches of 2 and 6 samples with losses 1 and 3: (2*1 + 6*3) / 8, not 3.0
    assert hist[0]["loss"] == pytest.approx(2.5) and seen[0]["loss"] == pytest.approx(2.5)
    assert hist[0]["val_loss"] == pytest.approx((10.0 + 20.0 + 2 * 60.0) / 4)
    # streaming metrics keep the last (cumulative) value
    assert hist[0]["logits_acc"] == pytest.approx(0.2) and hist[0]["val_logits_acc"] == pytest.approx(0.03)


def test_callbacks_monitoring_loss_act_on_the_epoch_mean():
    cb = pkg("callbacks")
    m = _stub_model([5.0, 1.0, 1.0, 5.0])       # epoch means 3.0 and 3.0; the last batches' values are 1.0 and 5.0

    class Opt:
        lr = 1.0
    m.optimizer = Opt()
    rl = cb.ReduceLROnPlateau(monitor="loss", factor=0.5, patience=1, epsilon=0.0)
    es = cb.EarlyStopping(monitor="loss", patience=1)
    m.fit_generator(_Gen([4]), 2, epochs=2, callbacks=[rl, es])
    assert rl.best == pytest.approx(3.0) and es.best == pytest.approx(3.0)
    assert m.stop_training      # no improvement of the MEAN in epoch 2 (the last batch alone would have said 1.0 -> 5.0 as well,
    #                             but best would be 1.0)


def test_stepwise_scheduler_leaves_beta_2_alone_by_default():
    cb = pkg("callbacks")

    class Opt:
        lr, beta_2 = 1.0, 0.999

    class M:
        optimizer = Opt()
    for apply, want in ((None, 0.999), (True, 0.99)):
        kw = {} if apply is None else {"apply_b2_schedule": apply}
        s = cb.StepwiseLearningRateScheduler(lr_schedule=lambda x: 0.5, b2_schedule=lambda x: 0.99, **kw)
        M.optimizer.beta_2 = 0.999
        s.set_model(M)
        s.on_batch_begin(0)
        assert M.optimizer.lr == 0.5 and M.optimizer.beta_2 == want


def test_schedule_interpreter_python2_arithmetic():
    S = pkg("schedules")
    assert S.parse_schedule("lambda x: x / 2")(5) == 2                 # int / int floors, as in the reference's interpreter
    assert S.parse_schedule("lambda x: float(x) / 2")(5) == 2.5
    assert S.parse_schedule("lambda x: x / 2.0")(5) == 2.5
    f = S.parse_schedule("lambda x: (x > 3 and 0.5) or 0.25")          # and / or return the deciding operand
    assert f(5) == 0.5 and f(1) == 0.25
    assert S.parse_schedule("lambda x: 0.99 if x < 40000 else 0.999")(50000) == 0.999


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sync_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import test_fit_loop_cpu as T
    from conftest import pkg as _pkg
    trainers, cb = _pkg("trainers"), _pkg("callbacks")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the ranks see different training and validation shards: rank 1's validation loss keeps falling, rank 0's stalls
    m = T._stub_model([1.0 + rank, 2.0 + rank], val_losses=[4.0, 4.0] if rank == 0 else [4.0, 1.0])

    class Opt:
        lr = 1.0
    m.optimizer = Opt()
    m.log_sync = trainers.EpochLogSync()
    rl = cb.ReduceLROnPlateau(monitor="val_loss", factor=0.5, patience=1, epsilon=0.0)
    es = cb.EarlyStopping(monitor="val_loss", patience=5)
    hist = m.fit_generator(T._Gen([2]), 1, epochs=4, callbacks=[rl, es], validation_data=T._Gen([3]), validation_steps=1)
    e = trainers.broadcast_object_from_rank0(7 if rank == 0 else 0)
    np.savez(os.path.join(out_dir, "r{}.npz".format(rank)), loss=[h["loss"] for h in hist], val=[h["val_loss"] for h in hist],
             lr=m.optimizer.lr, stop=m.stop_training, e=e)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_hand_identical_logs_to_their_callbacks(tmp_path):
    mp.start_processes(_sync_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["loss"], r1["loss"]) and np.array_equal(r0["val"], r1["val"])
    # epoch 0: train losses 1 (rank 0) and 2 (rank 1) over 2 samples each -> 1.5; validation 4 and 4 -> 4, then (4 + 1) / 2
    assert r0["loss"][0] == pytest.approx(1.5) and r0["val"][0] == pytest.approx(4.0) and r0["val"][1] == pytest.approx(2.5)
    assert float(r0["lr"]) == float(r1["lr"]) and bool(r0["stop"]) == bool(r1["stop"])
    assert int(r0["e"]) == 7 and int(r1["e"]) == 7
