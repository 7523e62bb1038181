"""Data-parallel supervised segmentation trainer (replaces the single-device loop of the reference's
src/trainers.py: TrainerBase.__init__ :170-282, _get_model_optimizer :1070-1125, SegmentationTrainer.train :2334-2376,
loss/metric wiring :2420-2460).

One process per GPU.  Every rank holds a full replica (flat parameter arena), runs forward/backward on its shard of the
global batch and the ranks average gradients with RCCL all-reduces over two buckets of the flat gradient arena; the
first bucket (late layers, ~95 % of the bytes) is reduced on the communicator's stream while the early layers' backward
is still running.  BatchNormalization statistics stay per replica (what any Keras data-parallel wrapper would do).
Evaluation shards the images and sums the int64 confusion matrices, so metrics do not depend on the rank count.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class GradientBucketReducer:
    """Sum-all-reduce of ranges of a flat gradient tensor, asynchronously on the process group's own stream.

    start(grad, (lo, hi)) enqueues the all-reduce of grad[lo:hi] behind everything already queued on the current
    stream; finish() makes the current stream wait for every pending bucket.  The optimizer divides by world_size
    (grad_scale), which reproduces the global-batch mean because the loss is a mean over equally sized shards
    (src/losses.py:263).  Works with any backend (nccl = RCCL on ROCm; gloo in CPU tests).
    """

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._pending = []
        self.bytes_reduced = 0
        self.buckets_started = 0

    def start(self, grad: torch.Tensor, rng: Tuple[int, int]):
        lo, hi = rng
        if hi <= lo:
            return
        view = grad[lo:hi]
        self.bytes_reduced += view.numel() * view.element_size()
        self.buckets_started += 1
        if self._host_staged(view):
            # gloo with device tensors (rehearsals of several ranks on one GPU only): the bucket travels through host
            # memory under this class's control - wait for the stream, copy out, reduce on the host, copy back.  gloo's own
            # device hand-off stalls for minutes when the all-reduce is queued behind device work of two processes that
            # share a GPU (reproduced with plain matrix products, scripts/gloo_speed_probe.py)
            torch.cuda.current_stream().synchronize()
            host = view.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(host)
            return
        if os.environ.get("SSMI_SYNC_ALLREDUCE"):       # debugging aid: blocking collectives
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return
        self._pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _host_staged(self, t: torch.Tensor) -> bool:
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def finish(self, after_each=None):
        """Makes the current stream wait for every pending bucket, in the order they were started.  after_each: called once
        per bucket after its wait has been queued (timing marks)."""
        for w in self._pending:
            w.wait()
            if after_each is not None:
                after_each(w)
        self._pending = []

    def broadcast_(self, tensors: Sequence[torch.Tensor], src: int = 0):
        for t in tensors:
            dist.broadcast(t, src=src, group=self.group)

    def sum_int64_(self, t: torch.Tensor):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class EpochLogSync:
    """Makes the numbers the callbacks act on identical on every rank: the per-epoch sums of the non-streaming outputs
    (value x batch size), the sample count and the int64 confusion matrix are summed over the process group before the
    engine turns them into epoch logs.  ReduceLROnPlateau, EarlyStopping and the checkpoint names then take the same
    decision on every rank (a rank that stopped or lowered its rate alone would leave the others waiting in the next
    gradient all-reduce, or let the replicas drift apart)."""

    def __init__(self, group=None):
        self.group = group

    def reduce(self, totals: Dict[str, float], seen: int, cm: Optional[np.ndarray]):
        keys = sorted(totals)
        t = torch.tensor([float(totals[k]) for k in keys] + [float(seen)], dtype=torch.float64)
        t = _all_reduce_host(t, self.group)
        out = {k: float(v) for k, v in zip(keys, t[:-1].tolist())}
        if cm is not None:
            cm = _all_reduce_host(torch.from_numpy(np.ascontiguousarray(cm, dtype=np.int64)), self.group).numpy()
        return out, int(round(float(t[-1]))), cm


def _all_reduce_host(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of a small host tensor over the group, whatever the backend (RCCL reduces device tensors only)."""
    if dist.get_backend(group) == "nccl":
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(d, op=dist.ReduceOp.SUM, group=group)
        return d.cpu()
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def broadcast_object_from_rank0(obj, group=None):
    """Rank 0's value of a small picklable object on every rank."""
    box = [obj]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


def all_reduce_evaluation(cm: torch.Tensor, loss_sum: float, count: int, group=None):
    """Sums a rank's int64 confusion matrix and its (loss * images, images) pair over the process group (no-op without
    one).  Returns (matrix as numpy int64, mean loss, images).  Integer sums are exact, so the matrix is the one a
    single process would have accumulated over all shards."""
    cm = cm.clone()
    tot = torch.tensor([loss_sum, float(count)], dtype=torch.float64, device=cm.device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(cm, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot = tot.cpu().numpy()
    return cm.cpu().numpy(), float(tot[0] / max(tot[1], 1.0)), int(tot[1])


def shard_range(global_batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous equal shards of the global batch (global batch 128 -> 8 x 16)."""
    if global_batch % world_size:
        raise ValueError("global batch {} is not divisible by world size {}".format(global_batch, world_size))
    per = global_batch // world_size
    return rank * per, (rank + 1) * per


class _StdoutToStderr:
    """Routes the PROCESS's stdout (file descriptor 1, i.e. C libraries too) to stderr for the duration of the block: RCCL
    prints a version banner to stdout when a communicator is created, which would otherwise land in front of a program's
    own machine-readable output."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def init_distributed_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    dev_index = local % ndev if ndev else 0     # more ranks than devices only happens in single-GPU rehearsals (gloo)
    # SSMI_FORCE_DP=1: a process group of ONE rank, so that the data-parallel step (gradient buckets on the collective
    # stream between the graph segments) can be exercised and timed on a single GPU
    force = bool(os.environ.get("SSMI_FORCE_DP")) and world == 1
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SSMI_DIST_BACKEND") or ("nccl" if ndev else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            with _StdoutToStderr():
                dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
                # the first collective creates the communicator (and prints RCCL's banner) if the eager path did not
                probe = torch.zeros(1, device=torch.device("cuda", dev_index))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
        else:
            if ndev:
                torch.cuda.set_device(dev_index)
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif ndev:
        torch.cuda.set_device(dev_index)
    local = dev_index
    return rank, local, world


class SegmentationTrainer(object):
    """Supervised segmentation trainer driven by the reference's JSON config keys
    (configs/segmentation/config-segmentation-supervised.json): input_shape, num_classes (from the material CSV in the
    reference; given directly here), num_labeled_per_batch, optimizer{name, learning_rate, momentum, decay},
    ignore_classes, use_class_weights / override_class_weights, random_seed, num_epochs."""

    def __init__(self, trainer_type, model_name, model_folder_name, config_file_path=None, config: Optional[dict] = None,
                 num_classes: int = 24, device=None):
        from . import models, losses, metrics
        from .keras_extensions import extended_optimizers as xo
        if trainer_type not in ("segmentation_supervised", "segmentation_supervised_mean_teacher",
                                "segmentation_semi_supervised_mean_teacher", "segmentation_semi_supervised_superpixel",
                                "segmentation_semi_supervised_mean_teacher_superpixel"):
            raise NotImplementedError("trainer type {} is not implemented (classification trainers)".format(trainer_type))
        self.trainer_type = trainer_type
        if config is None:
            with open(config_file_path) as f:
                config = json.load(f)
        self.config = config
        self.model_name = model_name
        self.model_folder_name = model_folder_name
        self.rank, self.local_rank, self.world_size = init_distributed_from_env()
        self.device = device or ("cuda:{}".format(self.local_rank))
        self.num_classes = num_classes
        self.random_seed = int(config.get("random_seed", 14874))
        self.ignore_classes = list(config.get("ignore_classes") or [])      # src/trainers.py:725-736: absent = none ignored
        self.num_epochs = int(config.get("num_epochs", 1))
        self.num_labeled_per_batch = int(config.get("num_labeled_per_batch", 16))
        self.num_unlabeled_per_batch = int(config.get("num_unlabeled_per_batch", 0)) if trainer_type.startswith("segmentation_semi") else 0
        self.mean_teacher_method_config = config.get("mean_teacher_params") or config.get("mean_teacher_method_config")
        self.teacher_model = None
        self._ema_fn = self._cons_fn = self._sp_fn = None
        self.superpixel_method_config = config.get("superpixel_params")
        cw = config.get("override_class_weights")
        if cw is None or not config.get("use_class_weights", True):
            cw = np.ones(num_classes, dtype=np.float32)
            for c in self.ignore_classes:
                cw[c] = 0.0   # src/trainers.py:2047-2053
        self.class_weights = np.asarray(cw, dtype=np.float32)
        T = losses.ModelLambdaLossType
        ltype = {(False, False): T.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY, (True, False): T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER,
                 (False, True): T.SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL,
                 (True, True): T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL}[(self.using_mean_teacher_method,
                                                                                          self.using_superpixel_method)]
        # "activation_dtype": "f32" (default) | "bf16": storage format of activations and their gradients in HBM
        mb = models.get_model(model_name, tuple(config.get("input_shape", [None, None, 3])), num_classes, ltype, device=self.device,
                              dtype=config.get("activation_dtype", "f32"))
        self.model_wrapper = mb
        self.model = mb.model
        self.model.seed = self.random_seed + self.rank   # independent dropout draws per replica
        self.model.graph.store.init_values(self.random_seed)
        oc = config.get("optimizer", {"name": "adam", "learning_rate": 5e-4})
        name = oc.get("name", "adam").lower()
        if name == "adam":
            opt = xo.Adam(lr=oc.get("learning_rate", 1e-3), decay=oc.get("decay", 0.0))
        elif name == "sgd":
            opt = xo.SGD(lr=oc.get("learning_rate", 0.01), momentum=oc.get("momentum", 0.0), decay=oc.get("decay", 0.0))
        else:
            raise ValueError("Unsupported optimizer name: {}".format(name))
        nu = self.num_unlabeled_per_batch      # src/trainers.py:2445-2448: the training metrics skip the unlabeled rows
        ic = self.ignore_classes
        self.model.compile(optimizer=opt, loss={'loss': losses.dummy_loss, 'logits': lambda _, y_pred: 0.0 * y_pred},
                           loss_weights={'loss': 1.0, 'logits': 0.0},
                           metrics={'logits': [metrics.segmentation_accuracy(nu, ic), metrics.segmentation_mean_iou(num_classes, nu, ic),
                                               metrics.segmentation_mean_per_class_accuracy(num_classes, nu, ic),
                                               metrics.segmentation_confusion_matrix(num_classes, nu, ic)]})
        self.reducer = None
        if self.world_size > 1 or (os.environ.get("SSMI_FORCE_DP") and dist.is_initialized()):
            self.reducer = GradientBucketReducer()
            st = self.model.graph.store
            self.reducer.broadcast_([st.theta, st.state], src=0)
            self.model.grad_sync = self.reducer
            self.model.log_sync = EpochLogSync(self.reducer.group)
        if self.using_mean_teacher_method:
            # src/trainers.py:1460-1520: the teacher has the student's topology without the lambda loss, starts from the
            # student's weights and is never trained; compile gives it the supervised loss + metrics for validation.
            # Every replica derives the same teacher: the students are identical after each all-reduced step.
            tb = models.get_model(model_name, tuple(config.get("input_shape", [None, None, 3])), num_classes,
                                  losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY, device=self.device)
            self.teacher_model_wrapper = tb
            self.teacher_model = tb.model
            self.teacher_model.seed = self.random_seed + 7919 + self.rank
            ts, ss = self.teacher_model.graph.store, self.model.graph.store
            ts.theta.copy_(ss.theta)
            ts.state.copy_(ss.state)
            self.teacher_model.compile(optimizer=xo.SGD(lr=0.0), loss={'loss': losses.dummy_loss, 'logits': lambda _, y_pred: 0.0 * y_pred},
                                       loss_weights={'loss': 1.0, 'logits': 0.0},
                                       metrics={'logits': [metrics.segmentation_accuracy(0, ic), metrics.segmentation_mean_iou(num_classes, 0, ic),
                                                           metrics.segmentation_mean_per_class_accuracy(num_classes, 0, ic),
                                                           metrics.segmentation_confusion_matrix(num_classes, 0, ic)]})

    @property
    def using_mean_teacher_method(self):
        return "mean_teacher" in self.trainer_type

    @property
    def using_superpixel_method(self):
        """src/trainers.py:2204-2207."""
        return self.trainer_type.endswith("superpixel")

    @property
    def superpixel_unlabeled_cost_coefficient_function(self):
        """src/trainers.py:1940-1947."""
        if self.using_superpixel_method and self._sp_fn is None:
            self._sp_fn = self._schedule((self.superpixel_method_config or {}).get('unlabeled_cost_coefficient_function', 1.0))
        return self._sp_fn

    @staticmethod
    def _schedule(spec):
        """The reference's configs give the schedules as Python lambda source ("lambda x: 0.99 if x < 40000 else
        0.999", src/trainers.py:1568-1583 evaluates them with eval); callables and plain numbers are accepted too.  The
        source is parsed and interpreted over a closed arithmetic vocabulary (schedules.parse_schedule), never eval'd."""
        from .schedules import parse_schedule
        return parse_schedule(spec)

    @property
    def ema_smoothing_coefficient_function(self):
        if self.using_mean_teacher_method and self._ema_fn is None:
            self._ema_fn = self._schedule((self.mean_teacher_method_config or {}).get('ema_smoothing_coefficient_function', 0.99))
        return self._ema_fn

    @property
    def consistency_cost_coefficient_function(self):
        if self.using_mean_teacher_method and self._cons_fn is None:
            self._cons_fn = self._schedule((self.mean_teacher_method_config or {}).get('consistency_cost_coefficient_function', 1.0))
        return self._cons_fn

    # hooks of the reference's engine (src/trainers.py:1314-1334)
    def modify_batch_data(self, step_index, x, y, validation=False):
        """src/trainers.py:1620-1664, 1822-1867: for a mean-teacher student the teacher's images (last input of a training
        batch, the student's own images when validating) are replaced by the teacher's logits, predicted with the
        training-phase layers active, followed by the per-row consistency coefficient; validation batches get zeros."""
        if not (self.using_mean_teacher_method or self.using_superpixel_method):
            return x, y
        x = list(x)
        img_batch = x[0]
        n = int(img_batch.shape[0])
        if self.using_mean_teacher_method:
            if not validation:
                teacher_img_batch, x = x[-1], x[:-1]
            else:
                teacher_img_batch = img_batch
        if len(x) != 4:
            raise ValueError("expected [images, labels, class_weights, num_unlabeled(, teacher_images)], got {} inputs".format(len(x)))
        if self.using_mean_teacher_method:
            if self.teacher_model is None:
                raise ValueError('Teacher model is not set, cannot run predictions')
            if validation:
                shape = tuple(int(v) for v in img_batch.shape[0:-1]) + (self.num_classes,)
                mt = torch.zeros(shape, dtype=torch.float32, device=self.device)
                coef = np.zeros([n, 1], dtype=np.float32)
            else:
                mt = self.teacher_model.predict_on_batch_device(teacher_img_batch, use_training_phase_layers=True)
                coef = np.ones([n, 1], dtype=np.float32) * float(self.consistency_cost_coefficient_function(step_index))
            x = x + [mt, coef]
        if self.using_superpixel_method:
            # src/trainers.py:2455-2472: zeros when validating, the scheduled coefficient otherwise
            sp = 0.0 if validation else float(self.superpixel_unlabeled_cost_coefficient_function(step_index))
            x = x + [np.ones([n, 1], dtype=np.float32) * sp]
        return x, y

    def on_batch_end(self, step_index):
        """src/trainers.py:1666-1707: exponential moving average of the student's weights into the teacher."""
        if not self.using_mean_teacher_method:
            return
        if self.teacher_model is None:
            raise ValueError('Teacher model is not set, cannot run EMA update')
        a = float(self.ema_smoothing_coefficient_function(step_index))
        self.teacher_model.ema_update_from(self.model, a)

    def evaluate_teacher(self, generator, steps) -> Dict[str, object]:
        """src/trainers.py:1734-1743: the teacher on the validation data (supervised loss + metrics), sharded and
        reduced like `evaluate`."""
        from . import metrics
        m = self.teacher_model
        m.reset_metrics()
        get = generator.__next__ if hasattr(generator, "__next__") else None
        loss_sum, count = 0.0, 0
        for b in range(steps):
            batch = get() if get else generator.get_batch(0, b)
            x = list(batch[0])[:4]
            outs = m.test_on_batch(x, batch[1])
            n = int(x[0].shape[0])
            loss_sum += float(outs[0]) * n
            count += n
        cm, loss, images = all_reduce_evaluation(m._cm, loss_sum, count, self.reducer.group if self.reducer else None)
        out = {"loss": loss, "images": images, "confusion_matrix": cm}
        out.update(metrics.metrics_from_confusion_matrix(cm, self.ignore_classes))
        return out

    def save_teacher_model_weights(self, file_path):
        """src/trainers.py:1770-1803 (the path template handling stays with the caller)."""
        if self.teacher_model is None:
            raise ValueError('Teacher model is not set, cannot save teacher model weights')
        if self.rank != 0:      # one writer per file: rank 0 (see save_model_weights)
            return
        os.makedirs(os.path.dirname(file_path) or ".", exist_ok=True)
        self.teacher_model.save(file_path, overwrite=True, include_optimizer=True)

    def on_epoch_end(self, epoch_index, step_index, logs):
        """src/trainers.py:1709-1768: the mean-teacher trainer saves the teacher beside the student every epoch."""
        tpl = (self.mean_teacher_method_config or {}).get('teacher_model_checkpoint_file_path')
        if self.using_mean_teacher_method and tpl and self.teacher_model is not None:
            val_loss = float(logs.get('val_loss', 0.0)) if logs else 0.0
            self.save_teacher_model_weights(self._populate_path_template(tpl, epoch=epoch_index, val_loss=val_loss))

    def _get_training_callbacks(self, initial_step=0):
        """The callbacks src/trainers.py:934-1068 builds from the configuration: model checkpoint (mandatory key
        'keras_model_checkpoint'), CSV log, early stopping, reduce-LR-on-plateau, stepwise LR scheduler, optimizer
        checkpoint.  TensorBoard is not available here.  Files are written by rank 0 only."""
        from . import callbacks as cb
        cfg = self.config
        writer = self.rank == 0
        out = []
        mc = cfg.get('keras_model_checkpoint')
        if not mc:
            raise ValueError('Could not find Keras ModelCheckpoint configuration with key keras_model_checkpoint - would be '
                             'unable to save model weights')
        out.append(cb.ModelCheckpoint(filepath=self._populate_path_template_partial(mc.get('checkpoint_file_path')),
                                      monitor=mc.get('monitor') or 'val_loss', verbose=mc.get('verbose') or 0,
                                      save_best_only=mc.get('save_best_only') or False,
                                      save_weights_only=mc.get('save_weights_only') or False, mode=mc.get('mode') or 'auto',
                                      period=mc.get('period') or 1, enabled=writer))
        csv_path = cfg.get('keras_csv_log_file_path')
        if csv_path is not None:
            csv_path = self._populate_path_template(csv_path)
            if os.path.exists(csv_path):      # never overwrite an earlier run's log
                import datetime
                csv_path = '{}-{:%Y-%m-%d-%H:%M}'.format(csv_path, datetime.datetime.now())
            out.append(cb.CSVLogger(csv_path, separator=',', append=False, enabled=writer))
        es = cfg.get('early_stopping')
        if es is not None:
            out.append(cb.EarlyStopping(monitor=es.get('monitor') or 'val_loss', min_delta=es.get('min_delta') or 0.0,
                                        patience=es.get('patience') or 2, verbose=es.get('verbose') or 0,
                                        mode=es.get('mode') or 'auto'))
        rl = cfg.get('reduce_lr_on_plateau')
        if rl is not None:
            out.append(cb.ReduceLROnPlateau(monitor=rl.get('monitor') or 'val_loss', factor=rl.get('factor') or 0.1,
                                            patience=rl.get('patience') or 10, min_lr=rl.get('min_lr') or 0,
                                            epsilon=rl.get('epsilon') or 0.0001, cooldown=rl.get('cooldown') or 0,
                                            verbose=rl.get('verbose') or 0))
        sw = cfg.get('stepwise_learning_rate_scheduler')
        if sw is not None:      # after reduce-on-plateau, so it overrides that callback's changes when both are present
            out.append(cb.StepwiseLearningRateScheduler(
                lr_schedule=self._schedule(sw.get('lr_schedule')) if sw.get('lr_schedule') is not None else None,
                b2_schedule=self._schedule(sw.get('b2_schedule')) if sw.get('b2_schedule') is not None else None,
                last_scheduled_step=sw.get('last_scheduled_step'), initial_step=initial_step, verbose=sw.get('verbose'),
                # the reference never applies b2_schedule (its own Adam fails the isinstance test in
                # stepwise_learning_rate_scheduler.py:62); "apply_b2_schedule": true in this block opts in
                apply_b2_schedule=bool(sw.get('apply_b2_schedule', False))))
        oc = cfg.get('optimizer_checkpoint_file_path')
        if oc is not None:
            out.append(cb.OptimizerCheckpoint(self._populate_path_template(oc), enabled=writer))
        return out

    def _populate_path_template_partial(self, path):
        """Fills {model_folder} and leaves the per-epoch fields ({epoch:02d}, {val_loss:.2f}) for the checkpoint callback."""
        return path.replace("{model_folder}", str(self.model_folder_name))

    def on_training_end(self):
        pass

    # ------------------------------------------------------------------------------------------------
    # checkpoints (src/trainers.py:1167-1238, 1367-1389); the container is .npz whatever the file is called
    # ------------------------------------------------------------------------------------------------
    def _populate_path_template(self, path, *args, **kwargs):
        """src/trainers.py:1136-1139: fills {model_folder} and any further fields the caller names (epoch, val_loss)."""
        kwargs.setdefault("log_folder_path", str(self.config.get("log_folder_path", "logs/{model_folder}")).replace(
            "{model_folder}", str(self.model_folder_name)))
        return path.format(model_folder=self.model_folder_name, *args, **kwargs)

    @staticmethod
    def _get_latest_weights_file_path(weights_directory_path, include_early_stop=False):
        """The checkpoint '<str>.<epoch>-<val_loss>.<str>' with the largest epoch in the directory (of the given file)."""
        folder = weights_directory_path if os.path.isdir(weights_directory_path) else os.path.dirname(weights_directory_path)
        if not os.path.isdir(folder):
            return None
        files = sorted(os.listdir(folder))
        if not include_early_stop:
            files = [f for f in files if 'early_stop' not in f]
        best, best_epoch = None, -100
        for f in files:
            if os.path.isfile(os.path.join(folder, f)) and ".hdf5" in f:
                try:
                    epoch = int(f.split('.')[1].rsplit('-', 1)[0])
                except (IndexError, ValueError):
                    continue
                if epoch > best_epoch:
                    best, best_epoch = f, epoch
        return os.path.join(folder, best) if best is not None else None

    def _load_latest_weights_for_model(self, model, weights_directory_path, include_early_stop=False) -> int:
        """Loads the newest checkpoint into `model` and returns the epoch to continue from (0 when there is none)."""
        path = self._get_latest_weights_file_path(weights_directory_path, include_early_stop=include_early_stop)
        if path is None:
            return 0
        try:
            model.load_weights(path)
            return int(os.path.basename(path).split('.')[1].rsplit('-', 1)[0]) + 1
        except Exception:      # src/trainers.py:1234-1236: any failure means "start from scratch"
            return 0

    def save_model_weights(self, epoch_index, val_loss, file_extension=''):
        template = (self.config.get('keras_model_checkpoint') or {}).get('checkpoint_file_path') or self.config.get('keras_model_checkpoint_file_path')
        if template is None:
            raise ValueError('the configuration names no checkpoint file path')
        file_path = self._populate_path_template(template, epoch=epoch_index, val_loss=val_loss) + file_extension
        # Data parallel: the weights are identical on every rank (same averaged gradients), the BatchNormalization moving
        # statistics are per replica; one writer per file, rank 0, so a checkpoint carries rank 0's moving statistics.
        if self.rank == 0:
            os.makedirs(os.path.dirname(file_path) or ".", exist_ok=True)
            self.model.save_weights(file_path, overwrite=True)
        return file_path

    def save_optimizer_settings(self, model, file_extension=''):
        file_path = self._populate_path_template(self.config['optimizer_checkpoint_file_path']) + file_extension
        if self.rank == 0:
            os.makedirs(os.path.dirname(file_path) or ".", exist_ok=True)
            with open(file_path, 'w') as f:
                f.write(json.dumps(model.optimizer.get_config()))
        return file_path

    def global_confusion_matrix(self) -> np.ndarray:
        cm = self.model._cm.clone()
        if self.reducer is not None:
            self.reducer.sum_int64_(cm)
        return cm.cpu().numpy()

    def evaluate(self, generator, steps) -> Dict[str, object]:
        """Sharded evaluation: every rank runs the evaluation loop of the reference
        (src/keras_extensions/extended_model.py:1223-1298) over ITS share of the batches, then the int64 confusion
        matrices and the (loss * images, images) pairs are summed over the ranks.  Integer sums do not depend on the
        order of addition, so the matrix and every metric derived from it are identical for any rank count."""
        from . import metrics
        m = self.model
        m.reset_metrics()
        get = generator.__next__ if hasattr(generator, "__next__") else None
        loss_sum, count = 0.0, 0
        for b in range(steps):
            batch = get() if get else generator.get_batch(0, b)
            x, y = self.modify_batch_data(b, batch[0], batch[1], validation=True)
            outs = m.test_on_batch(x, y)
            n = int(x[0].shape[0])
            loss_sum += float(outs[0]) * n
            count += n
        return self.reduce_evaluation(loss_sum, count)

    def reduce_evaluation(self, loss_sum: float, count: int) -> Dict[str, object]:
        from . import metrics
        cm, loss, images = all_reduce_evaluation(self.model._cm, loss_sum, count, self.reducer.group if self.reducer else None)
        out = {"loss": loss, "images": images, "confusion_matrix": cm}
        out.update(metrics.metrics_from_confusion_matrix(cm, self.ignore_classes))
        return out

    def train(self, generator, steps_per_epoch, validation_data=None, validation_steps=None):
        """src/trainers.py:2334-2376.  With 'keras_model_checkpoint' in the configuration the run resumes from the newest
        checkpoint of that folder ('continue_from_last_checkpoint', default on; weights, optimizer slots and iteration
        counter) and writes a checkpoint per epoch through the reference's callbacks; without the key (programmatic
        use, benchmarks) no callbacks run."""
        callbacks, initial_epoch = [], 0
        mc = self.config.get('keras_model_checkpoint')
        if mc:
            if self.config.get('continue_from_last_checkpoint', True):
                folder = os.path.dirname(self._populate_path_template_partial(mc.get('checkpoint_file_path')))
                path = self._get_latest_weights_file_path(folder) if os.path.isdir(folder) else None
                initial_epoch = self._load_latest_weights_for_model(self.model, folder) if path else 0
                if initial_epoch and self.config.get('continue_from_optimizer_checkpoint', True):
                    self.model_wrapper.load_optimizer_weights(path)
            if self.reducer is not None:
                # Every rank continues from rank 0's view: the epoch first (a rank that saw no checkpoint on its file system
                # must not train other epochs than the rest), then - unconditionally, so that no rank skips a collective -
                # weights, moving statistics, optimizer slots and the iteration counter.
                initial_epoch = int(broadcast_object_from_rank0(initial_epoch, self.reducer.group))
                st, opt = self.model.graph.store, self.model.optimizer
                self.reducer.broadcast_([st.theta, st.state] + list(opt.slot_tensors().values()) + [opt.step_block], src=0)
            callbacks = self._get_training_callbacks(initial_step=initial_epoch * steps_per_epoch)
        self.initial_epoch = initial_epoch
        return self.model.fit_generator(generator, steps_per_epoch, epochs=self.num_epochs, callbacks=callbacks,
                                        validation_data=validation_data, validation_steps=validation_steps,
                                        initial_epoch=initial_epoch, trainer=self, random_seed=self.random_seed)
