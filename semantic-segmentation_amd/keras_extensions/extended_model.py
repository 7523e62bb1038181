"""ExtendedModel: the training/evaluation engine behind `ModelBase.model`
(reference: src/keras_extensions/extended_model.py - compile :236-684, predict_on_batch :686-710,
fit_generator :713-1113, evaluate_generator :1116-1298, reset_metrics :139-157, write_cfm_to_file :115-137).

One `train_on_batch` of the reference is one tf.Session.run (feed -> forward -> loss -> gradients -> optimizer -> BN
updates -> streaming metric updates -> fetch).  Here the same step is a fixed sequence of libssmi.so kernel launches
on one HIP stream, captured once per input shape into HIP graphs and replayed:

    segment A : dropout tables, forward, fused loss+gradient+confusion matrix, backward of the late blocks
    (bucket 1 all-reduce starts here when data-parallel)
    segment B : backward of the early blocks
    (bucket 2 all-reduce)
    segment C : optimizer over the flat arena

Nothing in a step synchronises with the host unless the caller asks for the scalar results.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import Callable, Dict, List, Sequence

import numpy as np
import torch

from .._lib import lib
from ..engine import ENetGraph, _ptr, run_backward
from .. import metrics as metrics_mod


class LayerInfo:
    """Keras-style view of one layer: name, trainable flag, weight keys in Keras order."""

    def __init__(self, name, weight_keys):
        self.name, self.weight_keys, self.trainable = name, list(weight_keys), True


class _Marks:
    """Collects one timing event per call of append() (handed to the reducer, which calls it after each bucket wait)."""

    def __init__(self, mark):
        self._mark, self.events = mark, []

    def append(self, _=None):
        self.events.append(self._mark())


class ExtendedModel(object):

    def __init__(self, name: str, graph: ENetGraph, input_shape, num_classes: int, lambda_loss: bool,
                 mean_teacher: bool = False, superpixel: bool = False, classification: bool = False):
        self.name = name
        self.graph = graph
        self.input_shape = tuple(input_shape)
        self.num_classes = num_classes
        self.lambda_loss = lambda_loss
        # student of the mean-teacher method (src/models.py:188-190, src/losses.py:377-436): two more inputs, the
        # teacher's logits and the consistency-cost coefficient, and a loss over labeled + unlabeled rows
        self.mean_teacher = bool(mean_teacher and lambda_loss)
        self.input_names = ["images"] + (["labels", "class_weights", "num_unlabeled"] if lambda_loss else [])
        # superpixel cost over the unlabeled rows (src/models.py:191-196, src/losses.py:439-572): one more input, its coefficient
        self.superpixel = bool(superpixel and lambda_loss)
        # encoder-only model with the classification lambda loss (src/models.py:197-199, 706-729): labels and class weights
        # are [batch, classes] float tensors, the logits have rank 2
        self.classification = bool(classification and lambda_loss)
        self.act_dtype = graph.store.act_dtype
        # bfloat16 activations with the semi-supervised costs or the classification loss: those kernels take float32 logits,
        # so the (small) logits tensor is widened in front of them and the summed gradient rounded once behind them
        self._stage_logits = self.act_dtype != torch.float32 and (self.mean_teacher or self.superpixel or self.classification)
        self._logits_f32 = self._dlogits_f32 = None
        if self.mean_teacher:
            self.input_names += ["mt_predictions", "consistency_cost"]
        if self.superpixel:
            self.input_names += ["unlabeled_cost_coeff"]
        self._num_labeled = None
        self.mt_buf = self.coef_buf = self._cost_buf = None
        self.sp_mask_buf = self.sp_coef_buf = self._sp_cost_buf = self._sp_ws = None
        self.output_names = (["loss", "logits"] if lambda_loss else ["logits"])
        self.optimizer = None
        self.metrics = None
        self.metrics_names: List[str] = []
        self.metrics_cfm: List[str] = []
        self.metrics_streaming: List[str] = []
        self.metrics_hidden_from_progbar: List[str] = []
        self.metrics_excluded_from_callbacks: List[str] = []
        self._metric_fns: List[Callable] = []
        self.stop_training = False
        self._graphs: Dict[tuple, list] = {}
        self._seed = 14874
        self._unbiased_moving_variance = False
        self._keep_logits = False
        self.use_hip_graph = True
        self.grad_sync = None          # trainers.GradientBucketReducer when data-parallel
        self.log_sync = None           # trainers.EpochLogSync when data-parallel: epoch / validation logs summed over the ranks
        self._timed_steps = None       # list of per-step event records while segment_timing(True)
        self._step_marks = None
        self._bufs_shape = None
        self._noise_counter = None     # device int64: Philox step of predict_on_batch(use_training_phase_layers=True)
        self._cm = None
        self._ignore_bits = 0
        self._compiled = False
        # Keras layer list (weights grouped per layer, creation order)
        per_layer: "OrderedDict[str, list]" = OrderedDict()
        for key in graph.store.specs:
            lname = key.rsplit("/", 1)[0]
            per_layer.setdefault(lname, []).append(key)
        self.layers = [LayerInfo(n, ks) for n, ks in per_layer.items()]

    # values that are baked into a captured step as kernel arguments: changing them drops the captured graphs
    @property
    def seed(self):
        return self._seed

    @seed.setter
    def seed(self, v):
        if int(v) != self._seed:
            self._clear_graphs()
        self._seed = int(v)

    @property
    def keep_logits(self):
        """Training / evaluation steps of the ENets compute the logits inside the loss kernel and do not store them (nothing
        in the reference's step reads them: src/keras_extensions/extended_model.py:1002).  True has the same kernel write
        them to `graph.logits` as well (tests, debugging).  predict_on_batch always stores them."""
        return self._keep_logits

    @keep_logits.setter
    def keep_logits(self, v):
        if bool(v) != self._keep_logits:
            self._clear_graphs()
        self._keep_logits = bool(v)

    @property
    def unbiased_moving_variance(self):
        return self._unbiased_moving_variance

    @unbiased_moving_variance.setter
    def unbiased_moving_variance(self, v):
        if bool(v) != self._unbiased_moving_variance:
            self._clear_graphs()
        self._unbiased_moving_variance = bool(v)

    def _clear_graphs(self):
        """Drops every captured step, releasing the HIP graph objects."""
        graphs, self._graphs = self._graphs, {}
        if not graphs:
            return
        L = lib()
        torch.cuda.synchronize()      # a graph may still be executing
        for glist in graphs.values():
            for h in glist:
                L.graph_destroy(h)

    def close(self):
        self._clear_graphs()

    def __del__(self):
        try:
            self._clear_graphs()
        except Exception:      # interpreter shutdown: the library or torch may already be gone
            pass

    # ------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------
    def weight_names(self) -> List[str]:
        return list(self.graph.store.specs.keys())

    def get_weights(self) -> List[np.ndarray]:
        v = self.graph.store.views
        return [v[k].detach().cpu().numpy().copy() for k in self.graph.store.specs]

    def set_weights(self, weights: Sequence[np.ndarray]):
        specs = self.graph.store.specs
        if len(weights) != len(specs):
            raise ValueError('You called `set_weights(weights)` on model "{}" with a weight list of length {}, but the '
                             'model was expecting {} weights.'.format(self.name, len(weights), len(specs)))
        for (k, s), w in zip(specs.items(), weights):
            w = np.asarray(w)
            if tuple(w.shape) != s.shape:
                raise ValueError('Layer weight shape {} not compatible with provided weight shape {} ({})'.format(
                    s.shape, tuple(w.shape), k))
            self.graph.store.views[k].copy_(torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)))

    def get_weights_dict(self) -> "OrderedDict[str, np.ndarray]":
        return OrderedDict(zip(self.weight_names(), self.get_weights()))

    def set_weights_dict(self, d):
        self.set_weights([d[k] for k in self.weight_names()])

    def save_weights(self, filepath, overwrite=True):
        """Weights by Keras name ('layer/kernel', ...) in a .npz container (h5py is not available on this platform)."""
        if os.path.exists(filepath) and not overwrite:
            raise IOError("{} exists".format(filepath))
        with open(filepath, "wb") as f:      # a file object keeps the caller's name (checkpoint templates end in .hdf5)
            np.savez(f, **{k.replace("/", "::"): v for k, v in self.get_weights_dict().items()})

    def load_weights(self, filepath, by_name=False):
        with np.load(filepath, allow_pickle=False) as z:
            d = {k.replace("::", "/"): z[k] for k in z.files if not k.startswith("optimizer_weights::")}
        if by_name:
            cur = self.get_weights_dict()
            cur.update({k: v for k, v in d.items() if k in cur})
            d = cur
        self.set_weights_dict(d)

    def save(self, filepath, overwrite=True, include_optimizer=True):
        """Weights plus, like Keras Model.save, the optimizer state (slot arenas and the iteration counter) in one .npz
        container; `load_weights` reads the weights back, `ModelBase.load_optimizer_weights` the optimizer part."""
        if os.path.exists(filepath) and not overwrite:
            raise IOError("{} exists".format(filepath))
        d = {k.replace("/", "::"): v for k, v in self.get_weights_dict().items()}
        if include_optimizer and self.optimizer is not None:
            for k, t in self.optimizer.slot_tensors().items():
                d["optimizer_weights::" + k] = t.detach().cpu().numpy()
            d["optimizer_weights::iterations"] = np.array(self.optimizer.iterations, dtype=np.int64)
        with open(filepath, "wb") as f:
            np.savez(f, **d)

    def load_optimizer_weights(self, filepath) -> bool:
        """False when the file holds no (or incompatible) optimizer state, True after restoring it."""
        if self.optimizer is None:
            raise ValueError('The model must be compiled with an optimizer before loading optimizer weights')
        with np.load(filepath, allow_pickle=False) as z:
            names = [k for k in z.files if k.startswith("optimizer_weights::")]
            if not names:
                return False
            slots = self.optimizer.slot_tensors()
            for k, t in slots.items():
                key = "optimizer_weights::" + k
                if key not in z.files or tuple(z[key].shape) != tuple(t.shape):
                    return False
            for k, t in slots.items():
                t.copy_(torch.from_numpy(np.ascontiguousarray(z["optimizer_weights::" + k], dtype=np.float32)))
            if "optimizer_weights::iterations" in z.files:
                self.optimizer.set_iterations(int(z["optimizer_weights::iterations"]))
        return True

    def count_params(self):
        return sum(s.size for s in self.graph.store.specs.values())

    # ------------------------------------------------------------------------------------------------
    # compile
    # ------------------------------------------------------------------------------------------------
    def compile(self, optimizer, loss=None, metrics=None, loss_weights=None, **kwargs):
        """Keeps the reference's argument shapes.  `loss` / `loss_weights` dictionaries of the supervised trainer
        ({'loss': dummy_loss, 'logits': 0*y_pred}, {'loss': 1, 'logits': 0}; src/trainers.py:2436-2440) select the
        in-graph lambda loss, which is what the fused kernel computes; other combinations are rejected."""
        if not self.lambda_loss:
            raise NotImplementedError("compile() for training needs a model built with a lambda loss type")
        if isinstance(loss_weights, dict):
            if float(loss_weights.get("loss", 1.0)) != 1.0 or float(loss_weights.get("logits", 0.0)) != 0.0:
                raise NotImplementedError("only loss_weights {'loss': 1, 'logits': 0} are supported")
        self.optimizer = optimizer
        # layers frozen before compile() (ModelBase.transfer_weights) take no updates, as in Keras
        frozen = [k for l in self.layers if not l.trainable for k in l.weight_keys]
        optimizer.bind(self.graph.store, frozen)
        self.loss = loss
        self.loss_weights = loss_weights
        self.metrics = metrics
        fns: List[Callable] = []
        if isinstance(metrics, dict):
            for out_name, lst in metrics.items():
                if out_name != "logits":
                    raise ValueError("metrics can only be attached to the 'logits' output")
                fns.extend(lst)
        elif metrics:
            fns.extend(metrics)
        if fns and getattr(self, "classification", False):
            raise NotImplementedError("the classification models report the loss only; segmentation metrics need rank-4 logits")
        self._metric_fns = fns
        self.metrics_names = ["loss", "loss_loss", "logits_loss"]
        ignore = None
        nus = set()
        for fn in fns:
            nu_fn = getattr(fn, "num_unlabeled", 0)
            nus.add(int(nu_fn.reshape(-1)[0]) if torch.is_tensor(nu_fn) else int(nu_fn or 0))
            nm = "logits_" + fn.__name__
            self.metrics_names.append(nm)
            if getattr(fn, "streaming", False):
                self.metrics_streaming.append(nm)
            if getattr(fn, "cfm", False):
                self.metrics_cfm.append(nm)
            if getattr(fn, "hide_from_progbar", False):
                self.metrics_hidden_from_progbar.append(nm)
            if getattr(fn, "exclude_from_callbacks", False):
                self.metrics_excluded_from_callbacks.append(nm)
            ic = tuple(sorted(getattr(fn, "ignore_classes", []) or []))
            if ignore is not None and ic != ignore:
                raise NotImplementedError("all metrics must share the same ignore_classes")
            ignore = ic
        if len(nus) > 1:
            raise NotImplementedError("all metrics must share the same num_unlabeled")
        # src/metrics.py:255-259: in the training phase the metrics see the first batch - num_unlabeled rows only, in the
        # test phase every row
        self._metrics_num_unlabeled = nus.pop() if nus else 0
        self._ignore_classes = list(ignore or [])
        self._ignore_bits = metrics_mod.ignore_mask_bits(self._ignore_classes)
        dev = self.graph.device
        self._cm = torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64, device=dev) if fns else None
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self._bad = torch.zeros(1, dtype=torch.int32, device=dev)
        self._compiled = True
        self._clear_graphs()

    @property
    def using_cfm_metric(self):
        return len(self.metrics_cfm) > 0

    def reset_metrics(self):
        """extended_model.py:139-157: zero the streaming state at epoch boundaries."""
        if self._cm is not None:
            self._cm.zero_()

    def write_cfm_to_file(self, epoch, cfm_key, cfm, file_path=None, log_folder_path=None):
        if file_path is None:
            if log_folder_path is None:
                return
            file_path = os.path.join(log_folder_path, 'cfms/', '{}_{}.txt'.format(cfm_key, epoch))
        os.makedirs(os.path.dirname(file_path) or ".", exist_ok=True)
        np.savetxt(file_path, X=cfm)

    # ------------------------------------------------------------------------------------------------
    # batch staging
    # ------------------------------------------------------------------------------------------------
    def _ensure_buffers(self, n, h, w):
        if self._bufs_shape == (n, h, w):
            return
        dev = self.graph.device
        self.images_buf = torch.zeros(n, h, w, 3, dtype=torch.float32, device=dev)
        if self.classification:
            self.labels_buf = torch.zeros(n, self.num_classes, dtype=torch.float32, device=dev)
            self.weights_buf = torch.zeros(n, self.num_classes, dtype=torch.float32, device=dev)
        else:
            self.labels_buf = torch.zeros(n, h, w, dtype=torch.int32, device=dev)
            self.weights_buf = torch.zeros(n, h, w, dtype=torch.float32, device=dev)
        if self.mean_teacher:
            self.mt_buf = torch.zeros(n, h, w, self.num_classes, dtype=torch.float32, device=dev)
            self.coef_buf = torch.zeros(1, dtype=torch.float32, device=dev)
            self._cost_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        if self.superpixel:
            self.sp_mask_buf = torch.zeros(n, h, w, dtype=torch.int32, device=dev)
            self.sp_coef_buf = torch.zeros(1, dtype=torch.float32, device=dev)
            self._sp_cost_buf = torch.zeros(1, dtype=torch.float32, device=dev)
            self._sp_ws = None
        if self._stage_logits:      # float32 copies of the logits and their gradient for the loss kernels (allocated here, not
            shape = (n, self.num_classes) if self.classification else (n, h, w, self.num_classes)      # under a capture)
            self._logits_f32 = torch.empty(shape, dtype=torch.float32, device=dev)
            self._dlogits_f32 = torch.empty_like(self._logits_f32)
        self._num_labeled = n
        self.graph.plan(n, h, w)
        self._bufs_shape = (n, h, w)
        self._clear_graphs()

    @staticmethod
    def _to_dev(dst: torch.Tensor, src):
        if torch.is_tensor(src):
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        else:
            a = np.ascontiguousarray(src).reshape(tuple(dst.shape))
            dst.copy_(torch.from_numpy(a).to(dst.dtype), non_blocking=True)

    @staticmethod
    def _scalar(v) -> float:
        return float(v.reshape(-1)[0].item() if torch.is_tensor(v) else np.asarray(v).reshape(-1)[0])

    def set_batch(self, x):
        """Stage one batch [images, labels, class_weights, num_unlabeled] (numpy or torch) into the static device
        buffers the captured graphs read."""
        images = x[0]
        n, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        self._ensure_buffers(n, h, w)
        self._to_dev(self.images_buf, images)
        if self.lambda_loss and len(x) > 1:
            self._to_dev(self.labels_buf, x[1])
            self._to_dev(self.weights_buf, x[2])
            nu = x[3]
            nu = int(nu.reshape(-1)[0].item() if torch.is_tensor(nu) else np.asarray(nu).reshape(-1)[0])
            if self.mean_teacher or self.superpixel:
                if len(x) != len(self.input_names):
                    raise ValueError('Expected {} inputs {}, got: {}'.format(len(self.input_names), self.input_names, len(x)))
                if not 0 <= nu < n:
                    raise ValueError("num_unlabeled must lie in [0, batch): the classification cost needs a labeled row")
                self._num_labeled = n - nu
                if self.superpixel:
                    # the unlabeled rows of `labels` carry the superpixel masks (0 = border); keep them for the cost
                    self.sp_mask_buf.copy_(self.labels_buf)
                    self._to_dev(self.sp_coef_buf, np.asarray(self._scalar(x[-1]), dtype=np.float32).reshape(1))
                    if nu:
                        need = lib().superpixel_workspace_bytes(nu * h * w, self.num_classes)
                        if self._sp_ws is None or self._sp_ws.numel() * 4 < need:
                            self._sp_ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.graph.device)
                            self._clear_graphs()
                if nu:
                    # the metrics see zero labels on the unlabeled rows (src/generators.py:1764-1766)
                    self.labels_buf[n - nu:].zero_()
                if self.mean_teacher:
                    self._to_dev(self.mt_buf, x[4])
                    self.coef_buf.fill_(self._scalar(x[5]))      # src/losses.py:413: element 0 of the per-row vector
            elif nu != 0:
                # src/losses.py:327-329,369 slices only the weights; logits/labels keep the full batch, so the
                # reference's own graph is shape-inconsistent for num_unlabeled > 0 in the supervised loss
                raise ValueError("the supervised lambda loss requires num_unlabeled == 0")

    # ------------------------------------------------------------------------------------------------
    # step construction
    # ------------------------------------------------------------------------------------------------
    def _segments(self, training: bool) -> List[Callable[[int], None]]:
        g = self.graph
        L = g.store.L
        n, h, w = self._bufs_shape
        npix = n * h * w
        C_ = self.num_classes
        nu_m = getattr(self, "_metrics_num_unlabeled", 0)
        if training and not 0 <= nu_m < n:
            raise ValueError("metrics were built for {} unlabeled rows per batch, the batch has {} rows".format(nu_m, n))
        npix_m = (n - nu_m) * h * w if training else npix      # rows the confusion matrix covers

        L32 = lib()      # the float32 entry points (L is the view of the activation type)

        def staged_logits(st, want_grad):
            """float32 logits / gradient buffers for the loss kernels below: the graph's own (float32 storage) or widened
            copies (bfloat16 storage)."""
            if not self._stage_logits:
                return _ptr(g.logits), (_ptr(g.dlogits) if want_grad else None)
            L32.cast_bf16_to_f32(_ptr(g.logits), _ptr(self._logits_f32), g.logits.numel(), st)
            return _ptr(self._logits_f32), (_ptr(self._dlogits_f32) if want_grad else None)

        def unstage_gradient(st, want_grad):
            if self._stage_logits and want_grad:
                L32.cast_f32_to_bf16(_ptr(self._dlogits_f32), _ptr(g.dlogits), g.dlogits.numel(), st)

        def loss_call_semi(st, want_grad):
            # classification cost over the labeled rows, metrics over every row (src/losses.py:416-422), then the
            # mean-teacher consistency cost over every row (:428) and / or the superpixel cost over the unlabeled rows
            # (:484), each added to the loss and to the logits gradient
            nl = self._num_labeled
            npl = nl * h * w
            lg, dl = staged_logits(st, want_grad)
            L32.softmax_xent_fwd_bwd(lg, _ptr(self.labels_buf), _ptr(self.weights_buf), npl, C_, 1.0, dl,
                                     _ptr(self._loss_buf), _ptr(self._cm), npix_m, self._ignore_bits, None, _ptr(self._bad),
                                     _ptr(g.ws), g.ws.numel() * 4, st)
            if self.mean_teacher:
                L32.mt_consistency_fwd_bwd(lg, _ptr(self.mt_buf), npix, C_, _ptr(self.coef_buf), 1.0, npl, dl,
                                           _ptr(self._loss_buf), 1, _ptr(self._cost_buf), _ptr(g.ws), g.ws.numel() * 4, st)
            if self.superpixel:
                if nl < n:
                    off = npl * C_ * 4
                    L32.superpixel_cost_fwd_bwd(lg + off, _ptr(self.sp_mask_buf) + npl * 4, n - nl, h, w, C_,
                                                _ptr(self.sp_coef_buf), 1.0, 10e-7, (dl + off) if want_grad else None,
                                                1 if self.mean_teacher else 0, _ptr(self._loss_buf), 1, _ptr(self._sp_cost_buf),
                                                _ptr(self._sp_ws), self._sp_ws.numel() * 4, st)
                else:
                    L32.fill_f32(_ptr(self._sp_cost_buf), 0.0, 1, st)
            unstage_gradient(st, want_grad)

        def loss_call(st, want_grad):
            if self.classification:
                lg, dl = staged_logits(st, want_grad)
                L32.classification_xent_fwd_bwd(lg, _ptr(self.labels_buf), _ptr(self.weights_buf), n, C_, 10e-7, 1.0,
                                                dl, _ptr(self._loss_buf), st)
                unstage_gradient(st, want_grad)
                return
            if self.mean_teacher or self.superpixel:
                return loss_call_semi(st, want_grad)
            if getattr(g, "head_input", None) is not None:
                # the last layer runs inside the loss kernel; its output is stored only on request
                g.blocks[-1].xent_fused(g.head_input, self.labels_buf, self.weights_buf, npix, self._loss_buf, self._cm, npix_m,
                                        self._ignore_bits, self._bad, g.ws, st, want_grad, self.keep_logits)
                g.logits_valid = bool(self.keep_logits)
                return
            L.softmax_xent_fwd_bwd(_ptr(g.logits), _ptr(self.labels_buf), _ptr(self.weights_buf), npix, C_, 1.0,
                                   _ptr(g.dlogits) if want_grad else None, _ptr(self._loss_buf), _ptr(self._cm), npix_m,
                                   self._ignore_bits, None, _ptr(self._bad), _ptr(g.ws), g.ws.numel() * 4, st)

        # supervised segmentation loss: the last layer of the ENets runs inside the loss kernel (engine.LogitsHead.xent_fused)
        fuse_head = not (self.classification or self.mean_teacher or self.superpixel)

        if not training:
            def seg_eval(st):
                g.forward(self.images_buf, training=False, stream=st, fuse_head=fuse_head)
                loss_call(st, False)
            return [seg_eval]

        blocks = g.blocks
        cuts = self._cut_indices() if self.grad_sync is not None else []
        bounds = [len(blocks)] + cuts + [0]        # backward segment j covers blocks[bounds[j+1]:bounds[j]]
        carry = {}

        def seg_first(st):
            g.rotate_weights(st)      # data-gradient forms of the kernels, from the weights this step starts with
            if g.drop_groups:
                g.refresh_dropout(self.seed, self.optimizer.step_block, stream=st)
            g.forward(self.images_buf, training=True, update_moving=True, unbiased_moving=self.unbiased_moving_variance,
                      stream=st, fuse_head=fuse_head)
            loss_call(st, True)
            carry["d"] = g.dlogits
            seg_bwd(0)(st)

        def seg_bwd(j):
            def run(st):
                d = run_backward(reversed(blocks[bounds[j + 1]:bounds[j]]), carry["d"], g._cfg, last=bounds[j + 1] == 0)
                if self.grad_sync is not None or j == len(bounds) - 2:
                    g._cfg.join()       # a bucket must be complete before its all-reduce starts
                carry["d"] = d
            return run

        def seg_opt(st):
            # the hyper-parameter block is flushed by _run before the launch (a copy cannot be part of a captured graph)
            self.optimizer.apply(grad_scale=self._grad_scale(), stream=st, capturing=True)

        if self.grad_sync is None:
            # no collective between the segments: one graph for the whole step (also lets the weight gradients of the late
            # blocks keep running beside the early blocks' data-gradient chain instead of being joined in between)
            def seg_all(st):
                seg_first(st)
                seg_opt(st)
            return [seg_all]
        return [seg_first] + [seg_bwd(j) for j in range(1, len(bounds) - 1)] + [seg_opt]

    BUCKET_BYTES = 25 << 20      # gradient bytes per all-reduce for the wide models (SegNet 83 MB, U-Net 124 MB)
    MAX_BUCKETS = 8

    def _block_offsets(self):
        """Arena offset of every block's first parameter (None for parameter-free blocks), or None when the arena does not
        follow the block order (then the whole arena is one bucket)."""
        st = self.graph.store
        offs, last = [], -1
        for b in self.graph.blocks:
            o = [st.specs[c.name + "/kernel"].offset for c in b.convs() if c.name + "/kernel" in st.specs]
            o = min(o) if o else None
            if o is not None:
                if o <= last:
                    return None
                last = o
            offs.append(o)
        return offs

    def _cut_indices(self) -> List[int]:
        """Block indices (descending) in front of which the backward pass is cut so that the gradients behind the cut can be
        all-reduced while the rest of the backward pass runs.  ENets (1.46 MB of gradients): none by default, one bucket;
        with SSMI_DP_CUTS=1 one cut in front of en_bn_2.0 - the late blocks hold ~95 % of the parameters, the early blocks the
        high-resolution (slow) backward work that hides the first bucket's all-reduce.  Wide models: a cut whenever ~BUCKET_BYTES of gradients have become final, in the
        order the backward pass produces them (last layers first)."""
        blocks = self.graph.blocks
        offs = self._block_offsets()
        if offs is None:
            return []
        # (the ENet cut is opt-in, SSMI_DP_CUTS=1: with 1.46 MB of gradients the second graph boundary and the join of the
        # weight-gradient chain at the cut cost more - one-rank RCCL step 8.07 ms against 7.93 with ONE bucket behind the whole
        # backward pass, plain step 7.88 - than an all-reduce of that size can hide)
        for i, b in enumerate(blocks):
            if b.name == "en_bn_2.0" and offs[i] is not None and os.environ.get("SSMI_DP_CUTS") == "1":
                return [i]
        cuts, hi = [], self.graph.store.theta.numel()
        for i in range(len(blocks) - 1, 0, -1):
            if offs[i] is not None and (hi - offs[i]) * 4 >= self.BUCKET_BYTES and len(cuts) < self.MAX_BUCKETS - 1:
                cuts.append(i)
                hi = offs[i]
        return cuts

    def bucket_ranges(self):
        """[(lo, hi)] element ranges of the gradient arena, in the order they become final during backward."""
        offs = self._block_offsets()
        edges = [self.graph.store.theta.numel()] + [offs[i] for i in self._cut_indices()] + [0]
        return [(edges[j + 1], edges[j]) for j in range(len(edges) - 1)]

    def _grad_scale(self) -> float:
        return 1.0 / self.grad_sync.world_size if self.grad_sync is not None else 1.0

    def _run(self, training: bool):
        key = (training, self._bufs_shape, self.grad_sync is not None, self._num_labeled)
        stream = torch.cuda.current_stream()
        st = stream.cuda_stream
        if training:
            self.optimizer.flush_hyper(self._grad_scale())
        segs = self._segments(training)
        hooks = self._hooks(training)
        if not self.use_hip_graph or st == 0:
            for s, hk in zip(segs, hooks):
                s(st)
                hk()
            return
        L = lib()
        if key not in self._graphs:
            # nothing of this process may still be running on the device while the segments are recorded (parameter
            # broadcast, a previous step's collectives)
            torch.cuda.synchronize()
            glist = []
            for s in segs:
                L.graph_begin_capture(st)
                try:
                    s(st)
                finally:
                    h = C.c_void_p()
                    L.graph_end_capture(st, C.byref(h))
                glist.append(h)
            self._graphs[key] = glist
        timed = training and self.grad_sync is not None and self._timed_steps is not None
        if timed:
            def mark():
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                return e
            self._step_marks = _Marks(mark)
            seg_events = [mark()]
        for h, hk in zip(self._graphs[key], hooks):
            L.graph_launch(h, st)
            if timed:
                seg_events.append(mark())
            hk()
        if timed:
            self._timed_steps.append({"seg": seg_events, "waits": self._step_marks.events})
            self._step_marks = None

    def _hooks(self, training):
        """What the host does after each segment's launch: start that segment's bucket; after the last backward segment also
        make the stream wait for every bucket; nothing after the optimizer."""
        if not training or self.grad_sync is None:
            return [lambda: None]
        rs = self.bucket_ranges()
        gs = self.grad_sync
        grad = self.graph.store.grad
        hooks = [(lambda r=r: gs.start(grad, r)) for r in rs[:-1]]
        hooks.append(lambda: (gs.start(grad, rs[-1]), gs.finish(self._wait_marks())))
        hooks.append(lambda: None)
        return hooks

    # -- per-segment timing of data-parallel steps (HIP events on the step's stream) ---------------------------------------
    def _wait_marks(self):
        return self._step_marks.append if self._step_marks is not None else None

    def segment_timing(self, on: bool):
        """While on, every data-parallel training step records an event after each segment launch and after each bucket
        wait; segment_times_ms() turns them into per-step means."""
        self._timed_steps = [] if on else None

    def segment_times_ms(self):
        """{"segments": [ms per backward segment...], "bucket_waits": [ms the stream waited for each bucket after the last
        segment...], "optimizer": ms} averaged over the timed steps (synchronises)."""
        torch.cuda.synchronize()
        steps = self._timed_steps or []
        if not steps:
            return None
        nseg = len(steps[0]["seg"]) - 2
        out = {"segments": [0.0] * nseg, "bucket_waits": [0.0] * len(steps[0]["waits"]), "optimizer": 0.0}
        for rec in steps:
            ev = rec["seg"]
            for j in range(nseg):
                out["segments"][j] += ev[j].elapsed_time(ev[j + 1])
            prev = ev[nseg]
            for j, w in enumerate(rec["waits"]):
                out["bucket_waits"][j] += prev.elapsed_time(w)
                prev = w
            out["optimizer"] += prev.elapsed_time(ev[nseg + 1])
        k = float(len(steps))
        return {"segments": [v / k for v in out["segments"]], "bucket_waits": [v / k for v in out["bucket_waits"]],
                "optimizer": out["optimizer"] / k, "steps": len(steps)}

    # ------------------------------------------------------------------------------------------------
    # Keras-shaped batch API
    # ------------------------------------------------------------------------------------------------
    def _results(self):
        loss = float(self._loss_buf.item())
        bad = int(self._bad.item())
        if bad:
            self._bad.zero_()
            raise ValueError("{} label values outside [0, {})".format(bad, self.num_classes))
        out = [loss, loss, 0.0]
        if self._metric_fns:
            cm = self._cm.cpu().numpy()
            for fn in self._metric_fns:
                out.append(fn.value_from_cm(cm))
        return out

    def train_step_async(self):
        """One optimisation step on the staged batch; returns nothing and does not synchronise."""
        if not self._compiled:
            raise RuntimeError("compile() the model before training")
        self._run(True)

    def eval_step_async(self):
        """Forward + loss + confusion-matrix update on the staged batch (inference phase); does not synchronise."""
        if not self._compiled:
            raise RuntimeError("compile() the model before evaluating")
        self._run(False)

    def train_on_batch(self, x, y=None, sample_weight=None, class_weight=None):
        """Keras Model.train_on_batch: returns [total_loss, loss_loss, logits_loss, *metrics] (metrics_names order)."""
        if not self._compiled:
            raise RuntimeError("compile() the model before training")
        self.set_batch(x)
        self._run(True)
        return self._results()

    def test_on_batch(self, x, y=None, sample_weight=None):
        if not self._compiled:
            raise RuntimeError("compile() the model before evaluating")
        self.set_batch(x)
        self._run(False)
        return self._results()

    def predict_on_batch(self, x, use_training_phase_layers=False):
        """extended_model.py:686-710: logits for a batch; with use_training_phase_layers the BN/dropout layers run in
        their training phase (no moving-average updates, as in a Keras predict function)."""
        images = x[0] if isinstance(x, (list, tuple)) else x
        n, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        self._ensure_buffers(n, h, w)
        self._to_dev(self.images_buf, images)
        st = torch.cuda.current_stream().cuda_stream
        if use_training_phase_layers and self.graph.drop_groups:
            self._refresh_prediction_noise(st)
        self.graph.forward(self.images_buf, training=bool(use_training_phase_layers), update_moving=False, stream=st)
        return self.graph.logits.detach().float().cpu().numpy()

    def _refresh_prediction_noise(self, st):
        """Fresh SpatialDropout2D tables for a training-phase prediction: the Philox step is a device counter of this
        model's own that advances with every such call (the reference's teacher draws new noise for every prediction,
        src/trainers.py:1620-1664; the optimizer's step counter would stand still for a model that is never stepped)."""
        if self._noise_counter is None:
            self._noise_counter = torch.zeros(2, dtype=torch.int64, device=self.graph.device)
        self.graph.refresh_dropout(self.seed + 1, self._noise_counter, stream=st)
        self._noise_counter[0] += 1

    def predict_on_batch_device(self, x, use_training_phase_layers=False) -> torch.Tensor:
        """predict_on_batch without the host round trip: returns the graph's logits buffer (device, [n, h, w, classes]),
        valid until this model's next forward pass.  The mean-teacher trainer hands it straight to the student."""
        images = x[0] if isinstance(x, (list, tuple)) else x
        n, h, w = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        self._ensure_buffers(n, h, w)
        self._to_dev(self.images_buf, images)
        st = torch.cuda.current_stream().cuda_stream
        if use_training_phase_layers and self.graph.drop_groups:
            self._refresh_prediction_noise(st)
        self.graph.forward(self.images_buf, training=bool(use_training_phase_layers), update_moving=False, stream=st)
        lg = self.graph.logits
        return lg if lg.dtype == torch.float32 else lg.float()      # consumers of predictions take float32

    def ema_update_from(self, student: "ExtendedModel", a: float):
        """Teacher update of the mean-teacher method (src/trainers.py:1692-1706) over EVERY weight get_weights()
        lists - trainable ones and the BN moving statistics - as two launches over the flat arenas."""
        ts, ss = self.graph.store, student.graph.store
        if list(ts.specs.keys()) != list(ss.specs.keys()) or ts.theta.numel() != ss.theta.numel():
            raise ValueError('The weight arrays are not of the same length for the student and teacher: {} vs {}'.format(
                len(ts.specs), len(ss.specs)))
        L = lib()
        st = torch.cuda.current_stream().cuda_stream
        L.ema_update(_ptr(ts.theta), _ptr(ss.theta), float(a), ts.n_train, st)
        if ts.n_state:
            L.ema_update(_ptr(ts.state), _ptr(ss.state), float(a), ts.n_state, st)

    def consistency_cost(self) -> float:
        """The consistency term of the last step's loss (0 for models without the mean-teacher loss)."""
        return float(self._cost_buf.item()) if self._cost_buf is not None else 0.0

    def superpixel_cost(self) -> float:
        """The superpixel term of the last step's loss (0 for models without it or batches without unlabeled rows)."""
        return float(self._sp_cost_buf.item()) if self._sp_cost_buf is not None else 0.0

    def predict(self, x, batch_size=None, verbose=0):
        images = x[0] if isinstance(x, (list, tuple)) else x
        bs = batch_size or int(images.shape[0])
        outs = [self.predict_on_batch(images[i:i + bs]) for i in range(0, int(images.shape[0]), bs)]
        return np.concatenate(outs, axis=0)

    # ------------------------------------------------------------------------------------------------
    # generator loops (host side; extended_model.py:947-1093, 1223-1298)
    # ------------------------------------------------------------------------------------------------
    def _streaming_values(self, cm: np.ndarray) -> Dict[str, object]:
        """Values of the streaming metrics for a (possibly rank-summed) confusion matrix."""
        return {"logits_" + fn.__name__: fn.value_from_cm(cm) for fn in self._metric_fns}

    def _epoch_logs(self, totals: Dict[str, float], seen: int, last: Dict[str, object], prefix: str = "") -> Dict[str, object]:
        """Epoch values as the reference's ExtendedBaseLogger reports them (src/keras_extensions/extended_callbacks.py:35-63):
        streaming metrics keep their last (cumulative) value, every other output is the mean over the samples seen.  With a
        data-parallel `log_sync` (trainers.EpochLogSync) the sums, the sample count and the confusion matrix are summed over
        the ranks first, so that every rank hands identical numbers to its callbacks."""
        sync = getattr(self, "log_sync", None)
        cm = self._cm.cpu().numpy() if self._cm is not None else None
        if sync is not None:
            totals, seen, cm = sync.reduce(totals, seen, cm)
        logs = {}
        for nm in self.metrics_names:
            if nm in self.metrics_streaming:
                if nm in last:
                    logs[prefix + nm] = last[nm]
            elif nm in totals:
                logs[prefix + nm] = totals[nm] / max(seen, 1)
        if sync is not None and cm is not None:
            for nm, v in self._streaming_values(cm).items():
                if nm in self.metrics_streaming:
                    logs[prefix + nm] = v
        return logs

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None, initial_epoch=0, trainer=None, random_seed=None, **kwargs):
        from ..callbacks import CallbackList
        cbs = CallbackList(callbacks, self)      # Keras callback protocol (extended_model.py:933-1093 drives the same hooks)
        history = []
        step_index = initial_epoch * steps_per_epoch
        get = generator.__next__ if hasattr(generator, "__next__") else None
        self.stop_training = False
        cbs.on_train_begin()
        for epoch in range(initial_epoch, epochs):
            self.reset_metrics()
            cbs.on_epoch_begin(epoch)
            totals, seen, last = {}, 0, {}
            for b in range(steps_per_epoch):
                batch = get() if get else generator.get_batch(epoch, b)
                x, y = batch[0], batch[1]
                size = int(x[0].shape[0])
                cbs.on_batch_begin(b, {"batch": b, "size": size})
                if trainer is not None:
                    x, y = trainer.modify_batch_data(step_index, x, y, validation=False)
                outs = self.train_on_batch(x, y)
                last = dict(zip(self.metrics_names, outs))
                seen += size
                for nm, v in last.items():
                    if nm not in self.metrics_streaming:
                        totals[nm] = totals.get(nm, 0.0) + float(v) * size
                blogs = {k: v for k, v in last.items() if k not in self.metrics_excluded_from_callbacks}
                blogs.update(batch=b, size=size)
                cbs.on_batch_end(b, blogs)
                if trainer is not None:
                    trainer.on_batch_end(step_index)
                step_index += 1
            logs = self._epoch_logs(totals, seen, last)
            if validation_data is not None:
                val = self.evaluate_generator(validation_data, validation_steps, trainer=trainer, validation=True)
                logs.update({"val_" + k: v for k, v in zip(self.metrics_names, val)})
            history.append(logs)
            excluded = set(self.metrics_excluded_from_callbacks)
            cb_logs = {k: v for k, v in logs.items() if k not in excluded and (k[4:] if k.startswith("val_") else k) not in excluded}
            cbs.on_epoch_end(epoch, cb_logs)
            self.reset_metrics()
            if trainer is not None:
                trainer.on_epoch_end(epoch, step_index, logs)
            if self.stop_training:
                break
        cbs.on_train_end()
        if trainer is not None:
            trainer.on_training_end()
        return history

    def evaluate_generator(self, generator, steps, trainer=None, validation=False, **kwargs):
        """Aggregation of extended_model.py:1279-1298: streaming metrics take the last batch's (cumulative) value,
        the others a batch-size weighted average.  Data parallel (`log_sync` set): every rank evaluates its own batches and
        the sums / the confusion matrix are reduced over the ranks, so the result is the same on all of them."""
        self.reset_metrics()
        get = generator.__next__ if hasattr(generator, "__next__") else None
        totals, seen, last = {}, 0, {}
        for b in range(steps):
            batch = get() if get else generator.get_batch(0, b)
            x, y = batch[0], batch[1]
            if trainer is not None:
                x, y = trainer.modify_batch_data(b, x, y, validation=validation)
            last = dict(zip(self.metrics_names, self.test_on_batch(x, y)))
            size = int(x[0].shape[0])
            seen += size
            for nm, v in last.items():
                if nm not in self.metrics_streaming:
                    totals[nm] = totals.get(nm, 0.0) + float(v) * size
        logs = self._epoch_logs(totals, seen, last)
        return [logs.get(nm) for nm in self.metrics_names]
