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
    if isinstance(model_lambda_loss_type, ModelLambdaLossType):
        raise NotImplementedError('lambda loss type {} is outside the supervised hot path'.format(model_lambda_loss_type))
    raise ValueError('Unknown model lambda loss type: {}'.format(model_lambda_loss_type))


class ModelBase(object):
    """src/models.py:157-205."""

    def __init__(self, name, input_shape, num_classes, model_lambda_loss_type=ModelLambdaLossType.NONE, device="cuda"):
        self.name = name
        self.input_shape = tuple(input_shape)
        self.num_classes = num_classes
        self.model_lambda_loss_type = model_lambda_loss_type
        self.lambda_loss_function = get_lambda_loss_function(model_lambda_loss_type)
        if len(self.input_shape) != 3 or self.input_shape[2] != 3:
            raise ValueError('input_shape must be (H|None, W|None, 3), got {}'.format(self.input_shape))
        self.device = device
        graph = self._build_model()
        lambda_loss = model_lambda_loss_type == ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY
        if lambda_loss:
            self.name = self.name + '-CCE'
        self._model = ExtendedModel(self.name, graph, self.input_shape, num_classes, lambda_loss)

    @property
    def model(self):
        return self._model

    def _build_model(self):
        raise NotImplementedError('This method should be implemented in the class derived from ModelBase')


class ENetNaiveUpsampling(ModelBase):
    """src/models.py:1020-1319."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        if encoder_only:
            raise NotImplementedError('encoder-only classification heads are outside the segmentation hot path')
        super().__init__("ENet-Naive-Upsampling", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("naive", self.num_classes, self.device)


class ENetMaxUnpooling(ModelBase):
    """src/models.py:1326-1628."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        if encoder_only:
            raise NotImplementedError('encoder-only classification heads are outside the segmentation hot path')
        super().__init__("ENet-Max-Unpooling", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("unpool", self.num_classes, self.device)


class ENetNaiveUpsamplingEnhanced(ModelBase):
    """src/models.py:1642-1796."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        if encoder_only:
            raise NotImplementedError('encoder-only classification heads are outside the segmentation hot path')
        super().__init__("ENet-Naive-Upsampling-Enhanced", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return ENetGraph("enhanced", self.num_classes, self.device)


class UNetModel(ModelBase):
    """src/models.py:771-943."""

    def __init__(self, input_shape, num_classes, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        super().__init__("UNet", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return UNetSegNetGraph("unet", self.num_classes, self.device)


class SegNetBasicModel(ModelBase):
    """src/models.py:951-1012."""

    def __init__(self, input_shape, num_classes, encoder_only=False, model_lambda_loss_type=ModelLambdaLossType.NONE, **kw):
        if encoder_only:
            raise NotImplementedError('encoder-only classification heads are outside the segmentation hot path')
        super().__init__("SegNet-Basic", input_shape, num_classes, model_lambda_loss_type, **kw)

    def _build_model(self):
        return UNetSegNetGraph("segnet", self.num_classes, self.device)


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
    if model_name in _KNOWN:
        raise NotImplementedError('encoder-only classification heads ({}) are outside the segmentation hot path'.format(model_name))
    raise ValueError('Unknown model name: {}'.format(model_name))
