"""Model builders with the API of the reference's src/models.py: `get_model(model_name, input_shape, num_classes,
model_lambda_loss_type) -> ModelBase`, `ModelBase.model -> ExtendedModel`.

The builders produce the engine's fused block graphs (engine.ENetGraph, engine_unet.UNetSegNetGraph) instead of a
Keras functional graph; layer and
weight names, weight shapes and weight order are the reference's (see each builder's citation).  One deliberate
deviation: the naive decoder's second PReLU is named `{prefix}_prelu_2`; the reference names both decoder PReLUs
`{prefix}_prelu_1` (src/models.py:1231,1254), which Keras rejects as a duplicate layer name.
"""
from __future__ import annotations

from .engine import ENetGraph
from .engine_unet import UNetSegNetGraph
from .keras_extensions.extended_model import ExtendedModel
from .losses import ModelLambdaLossType
from . import losses

_KNOWN = ('unet', 'enet-naive-upsampling', 'enet-naive-upsampling-encoder-only', 'enet-max-unpooling',
          'enet-max-unpooling-encoder-only', 'segnet', 'segnet-encoder-only', 'enet-naive-upsampling-enhanced',
          'enet-naive-upsampling-enhanced-encoder-only')


def get_lambda_loss_function(model_lambda_loss_type):
    """src/models.py:131-149."""
    if model_lambda_loss_type == ModelLambdaLossType.NONE:
        return None
    if model_lambda_loss_type == ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY:
        return losses.segmentation_categorical_cross_entropy_lambda_loss
    if model_lambda_loss_type == ModelLambdaLossType.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER:
        return losses.segmentation_mean_teacher_lambda_loss
    if model_lambda_loss_type == ModelLambdaLossType.CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY:
        return losses.classification_categorical_crossentropy_lambda_loss
    if model_lambda_loss_type == ModelLambdaLossType.SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL:
        return losses.segmentation_superpixel_lambda_loss
    if model_lambda_loss_type == ModelLambdaLossType.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL:
        return losses.segmentation_mean_teacher_superpixel_lambda_loss
    if isinstance(model_lambda_loss_type, ModelLambdaLossType):
        raise NotImplementedError('lambda loss type {} is not implemented (the classification mean-teacher loss)'.format(model_lambda_loss_type))
    raise ValueError('Unknown model lambda loss type: {}'.format(model_lambda_loss_type))


class WeightTransferInformation(object):
    """src/models.py:29-47."""

    def __init__(self, num_transferred_layers, last_transferred_layer_name, num_frozen_layers, last_frozen_layer_name,
                 lr_scalers):
        self.num_transferred_layers = num_transferred_layers
        self.last_transferred_layer_name = last_transferred_layer_name
        self.num_frozen_layers = num_frozen_layers
        self.last_frozen_layer_name = last_frozen_layer_name
        self.lr_scalers = lr_scalers


def _norm_index(idx, n):
    if idx is None:
        return None
    return idx + n if idx < 0 else idx


class ModelBase(object):
    """src/models.py:157-205."""

    def __init__(self, name, input_shape, num_classes, model_lambda_loss_type=ModelLambdaLossType.NONE, device="cuda",
                 dtype="f32"):
        """`dtype` (no counterpart in the reference, which is float32 only): element type of the activation tensors in
        device memory, "f32" or "bf16".  Parameters, statistics, optimizer state and arithmetic are float32 either way."""
        import torch
        try:
            self.act_dtype = {"f32": torch.float32, "float32": torch.float32, torch.float32: torch.float32,
                              "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, torch.bfloat16: torch.bfloat16}[dtype]
        except KeyError:
            raise ValueError('Unknown activation dtype: {} (expected "f32" or "bf16")'.format(dtype))
        self.name = name
        self.input_shape = tuple(input_shape)
        self.num_classes = num_classes
        self.model_lambda_loss_type = model_lambda_loss_type
        self.lambda_loss_function = get_lambda_loss_function(model_lambda_loss_type)
        if len(self.input_shape) != 3 or self.input_shape[2] != 3:
            raise ValueError('input_shape must be (H|None, W|None, 3), got {}'.format(self.input_shape))
        self.device = device
        graph = self._build_model()
        T = ModelLambdaLossType
        mean_teacher = model_lambda_loss_type in (T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER,
                                                  T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL)
        superpixel = model_lambda_loss_type in (T.SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL,
                                                T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL)
        classification = model_lambda_loss_type == T.CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY
        if classification and not getattr(graph, "encoder_only", False):
            raise ValueError('the classification lambda loss needs an encoder-only model (logits of rank 2)')
        if getattr(graph, "encoder_only", False) and model_lambda_loss_type not in (T.NONE, T.CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY):
            raise ValueError('encoder-only models take the classification lambda loss, got {}'.format(model_lambda_loss_type))
        lambda_loss = (mean_teacher or superpixel or classification
                       or model_lambda_loss_type == T.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY)
        if lambda_loss:      # src/models.py:185-199
            self.name = self.name + (('-MT' if mean_teacher else '') + ('-SP' if superpixel else '') or '-CCE')
        self._model = ExtendedModel(self.name, graph, self.input_shape, num_classes, lambda_loss, mean_teacher=mean_teacher,
                                    superpixel=superpixel, classification=classification)

    @property
    def model(self):
        return self._model

    def _build_model(self):
        raise NotImplementedError('This method should be implemented in the class derived from ModelBase')

    def transfer_weights(self, from_model, from_layer_index, to_layer_index, freeze_from_layer_index=None,
                         freeze_to_layer_index=None, scale_lr_from_layer_index=None, scale_lr_to_layer_index=None,
                         scale_lr_factor=None):
        """src/models.py:211-301, kept to the letter: the index range slices the get_weights() LIST of both models
        ("assumes indexing is the same for both models"), freezing marks LAYERS [freeze_from, freeze_to) untrainable
        (call compile() afterwards), and the returned lr_scalers are keyed by the indices given."""
        n = len(from_model.layers)
        from_layer_index, to_layer_index = _norm_index(from_layer_index, n), _norm_index(to_layer_index, n)
        freeze_from_layer_index, freeze_to_layer_index = _norm_index(freeze_from_layer_index, n), _norm_index(freeze_to_layer_index, n)
        scale_lr_from_layer_index, scale_lr_to_layer_index = _norm_index(scale_lr_from_layer_index, n), _norm_index(scale_lr_to_layer_index, n)
        freezing = freeze_from_layer_index is not None and freeze_to_layer_index is not None
        scaling = scale_lr_from_layer_index is not None and scale_lr_to_layer_index is not None
        assert from_layer_index < to_layer_index
        assert not (freezing and freeze_from_layer_index > freeze_to_layer_index)
        assert not (scaling and scale_lr_from_layer_index > scale_lr_to_layer_index)
        assert not (scaling and scale_lr_factor is None)
        to_w = self._model.get_weights()
        from_w = from_model.get_weights()
        to_w[from_layer_index:to_layer_index] = from_w[from_layer_index:to_layer_index]
        self._model.set_weights(to_w)
        num_frozen = 0
        if freezing:
            for i in range(freeze_from_layer_index, freeze_to_layer_index):
                self._model.layers[i].trainable = False
                num_frozen += 1
        lr_scalers = {}
        if scaling:
            for i in range(scale_lr_from_layer_index, scale_lr_to_layer_index):
                lr_scalers[i] = scale_lr_factor
        return WeightTransferInformation(to_layer_index - from_layer_index, from_model.layers[to_layer_index - 1].name,
                                         num_frozen, from_model.layers[freeze_to_layer_index - 1].name if freezing else None,
                                         lr_scalers)

    def transfer_weights_from_file(self, filepath, from_layer_index, to_layer_index, freeze_from_layer_index=None,
                                   freeze_to_layer_index=None, scale_lr_from_layer_index=None,
                                   scale_lr_to_layer_index=None, scale_lr_factor=None):
        """Topological (order-based) transfer from a weight container (src/models.py:303-516): the layers WITH weights
        of the file and of this model are paired by position, layers [from, to) are copied (shapes must agree),
        freezing / learning-rate scaling as in transfer_weights but counted over the layers with weights; the
        lr_scalers index the model's trainable weights of the scaled layers.  The container is the .npz written by
        ExtendedModel.save_weights / save (h5py is not available on this platform)."""
        import numpy as np
        from collections import OrderedDict
        with np.load(filepath, allow_pickle=False) as z:
            file_layers: "OrderedDict[str, list]" = OrderedDict()
            for k in z.files:
                if k.startswith("optimizer_weights::"):
                    continue
                lname, _ = k.rsplit("::", 1)
                file_layers.setdefault(lname, []).append(np.asarray(z[k]))
        model_layers = [l for l in self._model.layers if l.weight_keys]
        n = len(file_layers)
        from_layer_index, to_layer_index = _norm_index(from_layer_index, n), _norm_index(to_layer_index, n)
        freeze_from_layer_index, freeze_to_layer_index = _norm_index(freeze_from_layer_index, n), _norm_index(freeze_to_layer_index, n)
        scale_lr_from_layer_index, scale_lr_to_layer_index = _norm_index(scale_lr_from_layer_index, n), _norm_index(scale_lr_to_layer_index, n)
        freezing = freeze_from_layer_index is not None and freeze_to_layer_index is not None
        scaling = scale_lr_from_layer_index is not None and scale_lr_to_layer_index is not None
        assert from_layer_index < to_layer_index
        assert not (scaling and scale_lr_factor is None)
        if to_layer_index > min(n, len(model_layers)):
            raise ValueError('Cannot transfer layers up to index {}: the file holds {} layers with weights, the model {}'
                             .format(to_layer_index, n, len(model_layers)))
        names = list(file_layers.keys())
        cur = self._model.get_weights_dict()
        for i in range(from_layer_index, to_layer_index):
            layer, vals = model_layers[i], file_layers[names[i]]
            if len(vals) != len(layer.weight_keys):
                raise ValueError('Layer #{} (named "{}") expects {} weight(s), but the saved weights have {} element(s)'
                                 .format(i, layer.name, len(layer.weight_keys), len(vals)))
            for key, v in zip(layer.weight_keys, vals):
                if tuple(v.shape) != tuple(cur[key].shape):
                    raise ValueError('Layer #{} (named "{}"): weight shape {} not compatible with provided shape {}'
                                     .format(i, layer.name, cur[key].shape, v.shape))
                cur[key] = v
        self._model.set_weights_dict(cur)
        num_frozen = 0
        if freezing:
            for i in range(freeze_from_layer_index, freeze_to_layer_index):
                model_layers[i].trainable = False
                num_frozen += 1
        lr_scalers = {}
        if scaling:
            store = self._model.graph.store
            trainable = [k for l in model_layers if l.trainable for k in l.weight_keys if store.specs[k].trainable]
            pos = {k: i for i, k in enumerate(trainable)}
            for i in range(scale_lr_from_layer_index, scale_lr_to_layer_index):
                for k in model_layers[i].weight_keys:
                    if k in pos:
                        lr_scalers[pos[k]] = scale_lr_factor
        return WeightTransferInformation(to_layer_index - from_layer_index, model_layers[to_layer_index - 1].name, num_frozen,
                                         model_layers[freeze_to_layer_index - 1].name if freezing else None, lr_scalers)

    def load_optimizer_weights(self, weights_file_path):
        """src/models.py:518-564."""
        if self.model is None:
            raise ValueError('The internal model must be built before loading optimizer weights')
        if self.model.optimizer is None:
            raise ValueError('The model must be compiled with an optimizer before loading optimizer weights')
        return self.model.load_optimizer_weights(weights_file_path)


class ENetNaiveUpsampling(ModelBase):
    """src/models.py:1020-1319."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        self.encoder_only = encoder_only
        super().__init__("ENet-Naive-Upsampling" if not encoder_only else "ENet-Naive-Upsampling-Encoder-Only", input_shape, num_classes,
                         model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("naive", self.num_classes, self.device, encoder_only=self.encoder_only, act_dtype=self.act_dtype)


class ENetMaxUnpooling(ModelBase):
    """src/models.py:1326-1628."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        self.encoder_only = encoder_only
        super().__init__("ENet-Max-Unpooling" if not encoder_only else "ENet-Max-Unpooling-Encoder-Only", input_shape, num_classes,
                         model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("unpool", self.num_classes, self.device, encoder_only=self.encoder_only, act_dtype=self.act_dtype)


class ENetNaiveUpsamplingEnhanced(ModelBase):
    """src/models.py:1642-1796."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        self.encoder_only = encoder_only
        super().__init__("ENet-Naive-Upsampling-Enhanced" if not encoder_only else "ENet-Naive-Upsampling-Enhanced-Encoder-Only", input_shape, num_classes,
                         model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("enhanced", self.num_classes, self.device, encoder_only=self.encoder_only, act_dtype=self.act_dtype)


class UNetModel(ModelBase):
    """src/models.py:771-943."""

    def __init__(self, input_shape, num_classes, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        super().__init__("UNet", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return UNetSegNetGraph("unet", self.num_classes, self.device, act_dtype=self.act_dtype)


class SegNetBasicModel(ModelBase):
    """src/models.py:951-1012."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        self.encoder_only = encoder_only
        super().__init__("SegNet-Basic" if not encoder_only else "SegNet-Basic-Encoder-Only", input_shape, num_classes,
                         model_lambda_loss_type, **kw)

    def _build_model(self):
        return UNetSegNetGraph("segnet", self.num_classes, self.device, encoder_only=self.encoder_only, act_dtype=self.act_dtype)


def get_model(model_name, input_shape, num_classes, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
    """src/models.py:54-128."""
    if model_name == 'enet-naive-upsampling':
        return ENetNaiveUpsampling(input_shape, num_classes, False, model_lambda_loss_type, **kw)
    if model_name == 'enet-max-unpooling':
        return ENetMaxUnpooling(input_shape, num_classes, False, model_lambda_loss_type, **kw)
    if model_name == 'enet-naive-upsampling-enhanced':
        return ENetNaiveUpsamplingEnhanced(input_shape, num_classes, False, model_lambda_loss_type, **kw)
    if model_name == 'unet':
        return UNetModel(input_shape, num_classes, model_lambda_loss_type, **kw)
    if model_name == 'segnet':
        return SegNetBasicModel(input_shape, num_classes, False, model_lambda_loss_type, **kw)
    if model_name == 'enet-naive-upsampling-encoder-only':
        return ENetNaiveUpsampling(input_shape, num_classes, True, model_lambda_loss_type, **kw)
    if model_name == 'enet-max-unpooling-encoder-only':
        return ENetMaxUnpooling(input_shape, num_classes, True, model_lambda_loss_type, **kw)
    if model_name == 'enet-naive-upsampling-enhanced-encoder-only':
        return ENetNaiveUpsamplingEnhanced(input_shape, num_classes, True, model_lambda_loss_type, **kw)
    if model_name == 'segnet-encoder-only':
        return SegNetBasicModel(input_shape, num_classes, True, model_lambda_loss_type, **kw)
    raise ValueError('Unknown model name: {}'.format(model_name))
