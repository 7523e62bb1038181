"""Full-image inference with the call shapes of the reference's src/predict.py.

The reference loads one image, normalises it, pads it so that both sides can be halved `div2_constraint` times, runs the
network on the whole frame, crops the padding off the activations, weights the background class, averages the
predictions of the ensemble's scales and ranks the classes per pixel.  Here everything between the uint8 image and the
ranked classes runs on the device: normalise + pad (`ssmi_predict_prepare`), the network in its inference phase, crop +
background weight + ensemble sum (`ssmi_predict_accumulate`), mean + top-k (`ssmi_predict_topk`).  Two steps of the
ensembling path stay host-side as in the reference: PIL's bicubic resize of the input and scipy's spline zoom that
brings a scaled prediction back to the original size.  Plotting and the dense-CRF post-process of the reference need
matplotlib / pydensecrf and are not part of this module.
"""
from __future__ import annotations

import math
from typing import List

import numpy as np
import torch

from ._lib import lib
from .generators import _normalised_range

ENSEMBLING_SCALE_FACTORS = [1.0 / math.sqrt(2.0), 1.0, math.sqrt(2)]      # src/predict.py:454


def np_softmax(X, theta=1.0, axis=None):
    """src/predict.py:37-79 (host numpy, used for the CRF unary in the reference)."""
    y = np.atleast_2d(X)
    if axis is None:
        axis = next(j[0] for j in enumerate(y.shape) if j[1] > 1)
    y = y * float(theta)
    y = y - np.expand_dims(np.max(y, axis=axis), axis)
    y = np.exp(y)
    p = y / np.expand_dims(np.sum(y, axis=axis), axis)
    return p.flatten() if X.ndim == 1 else p


def count_trailing_zeroes(num: int) -> int:
    """src/utils/dataset_utils.py:1364-1381."""
    if num <= 0:
        raise ValueError("count_trailing_zeroes needs a positive number, got {}".format(num))
    z = 0
    while ((num >> z) & 1) == 0:
        z += 1
    return z


def get_closest_higher_number_with_n_trailing_zeroes(num: int, n: int) -> int:
    """src/utils/dataset_utils.py:1338-1361."""
    if count_trailing_zeroes(num) >= n:
        return num
    unit = 1 << n
    if num < unit:
        return unit
    return num + (unit - num % unit)


def div2_padding(height: int, width: int, div2_constraint: int):
    """Padding of src/predict.py:164-178: (padded_h, padded_w, v_before, v_after, h_before, h_after); the odd pixel goes after."""
    ph = get_closest_higher_number_with_n_trailing_zeroes(height, div2_constraint)
    pw = get_closest_higher_number_with_n_trailing_zeroes(width, div2_constraint)
    vd, hd = max(0, ph - height), max(0, pw - width)
    return ph, pw, vd // 2, vd // 2 + vd % 2, hd // 2, hd // 2 + hd % 2


class DataSetInformation(object):
    """The four statistics src/predict.py:100-101 reads from the data set information file."""

    def __init__(self, per_channel_mean, per_channel_stddev, labeled_per_channel_mean=None, labeled_per_channel_stddev=None):
        self.per_channel_mean = per_channel_mean
        self.per_channel_stddev = per_channel_stddev
        self.labeled_per_channel_mean = labeled_per_channel_mean if labeled_per_channel_mean is not None else per_channel_mean
        self.labeled_per_channel_stddev = labeled_per_channel_stddev if labeled_per_channel_stddev is not None else per_channel_stddev


def pil_resize_image_to_sdim(pil_image, sdim, interp='nearest'):
    """src/predict.py:128-139."""
    func = {'nearest': 0, 'lanczos': 1, 'bilinear': 2, 'bicubic': 3, 'cubic': 3}
    scale_factor = float(sdim) / float(min(pil_image.width, pil_image.height))
    if scale_factor == 1.0:
        return pil_image
    return pil_image.resize((int(round(scale_factor * pil_image.width)), int(round(scale_factor * pil_image.height))),
                            resample=func[interp])


def _scale_u8(np_u8: np.ndarray, sfactor: float) -> np.ndarray:
    """src/utils/image_utils.py:1359-1370 with interp='bicubic': through an 8-bit PIL image, like the reference."""
    from PIL import Image
    th, tw = int(round(sfactor * np_u8.shape[0])), int(round(sfactor * np_u8.shape[1]))
    return np.array(Image.fromarray(np_u8, "RGB").resize((tw, th), resample=3), dtype=np.uint8)


class PredictionImage(object):
    """src/predict.py:86-125.  `image` is a path (loaded with PIL) or an HxWx3 array with values in [0, 255]; the
    normalised + padded network input is produced on the device by `device_input()` instead of in the constructor."""

    def __init__(self, image, data_set_information, sdim=None, scale_factor=1.0, div2_constraint=4, labeled_only=False):
        if isinstance(image, str):
            from PIL import Image
            pil = Image.open(image).convert("RGB")
            if sdim is not None:
                pil = pil_resize_image_to_sdim(pil, sdim, interp='bicubic')
            arr = np.asarray(pil, dtype=np.uint8)
            self.file_name = image.rsplit("/", 1)[-1]
        else:
            arr = np.asarray(image)
            if arr.ndim != 3 or arr.shape[2] != 3:
                raise ValueError("image must be HxWx3, got {}".format(arr.shape))
            if arr.min() < 0 or arr.max() > 255:
                raise ValueError('Image values are not in range [0, 255], got [{}, {}]'.format(arr.min(), arr.max()))
            if arr.dtype != np.uint8:
                if not np.array_equal(arr, np.round(arr)):
                    raise ValueError("the device path takes 8-bit images (integer values in [0, 255])")
                arr = arr.astype(np.uint8)
            if sdim is not None:
                from PIL import Image
                arr = np.asarray(pil_resize_image_to_sdim(Image.fromarray(arr, "RGB"), sdim, interp='bicubic'), dtype=np.uint8)
            self.file_name = None
        dsi = data_set_information
        mean = dsi.labeled_per_channel_mean if labeled_only else dsi.per_channel_mean
        std = dsi.labeled_per_channel_stddev if labeled_only else dsi.per_channel_stddev
        self.mean = _normalised_range(mean, "Per channel mean")
        self.std = _normalised_range(std, "Per-channel stddev")
        # the fill colour of the padding is the RAW per-channel mean of the whole data set (src/predict.py:124)
        self.cval = np.array(dsi.per_channel_mean, dtype=np.float32) if dsi.per_channel_mean is not None else np.zeros(3, np.float32)
        self.original_shape = arr.shape[:2]
        self.scale_factor = scale_factor
        self.scaled = scale_factor != 1.0
        if self.scaled:
            arr = _scale_u8(arr, scale_factor)
        self.np_image_u8 = np.ascontiguousarray(arr)
        h, w = arr.shape[:2]
        self.unpadded_shape = (h, w)
        self.v_pad_before = self.v_pad_after = self.h_pad_before = self.h_pad_after = 0
        self.padded = count_trailing_zeroes(h) < div2_constraint or count_trailing_zeroes(w) < div2_constraint
        ph, pw = h, w
        if self.padded:
            ph, pw, self.v_pad_before, self.v_pad_after, self.h_pad_before, self.h_pad_after = div2_padding(h, w, div2_constraint)
        self.padded_shape = (ph, pw)

    def device_input(self, device="cuda") -> torch.Tensor:
        """The network input [1, padded_h, padded_w, 3] (float32, device)."""
        L = lib()
        dev = torch.device(device)
        img = torch.from_numpy(self.np_image_u8).to(dev)
        mean = torch.from_numpy(self.mean).to(dev) if self.mean is not None else None
        std = torch.from_numpy(self.std).to(dev) if self.std is not None else None
        cval = torch.from_numpy(self.cval).to(dev)
        ph, pw = self.padded_shape
        h, w = self.unpadded_shape
        X = torch.empty(1, ph, pw, 3, dtype=torch.float32, device=dev)
        L.predict_prepare(img.data_ptr(), None if mean is None else mean.data_ptr(), None if std is None else std.data_ptr(),
                          h, w, self.v_pad_before, self.h_pad_before, ph, pw, cval.data_ptr(), X.data_ptr(),
                          torch.cuda.current_stream(dev).cuda_stream)
        self._keep = (img, mean, std, cval)     # alive until the stream has consumed them
        return X


def _zoom_back(pred: np.ndarray, scale_factor: float, original_shape) -> np.ndarray:
    """src/predict.py:495-516: spline zoom back to the original size and the reference's off-by-one repair."""
    from scipy import ndimage
    sf = 1.0 / scale_factor
    out = ndimage.zoom(pred, zoom=(sf, sf, 1.0), order=3, mode='constant', cval=-1.0).astype(pred.dtype)
    if out.shape[0] > original_shape[0]:
        out = out[out.shape[0] - original_shape[0]:, :, :]
    if out.shape[1] > original_shape[1]:
        out = out[:, out.shape[1] - original_shape[1]:, :]
    if out.shape[0] < original_shape[0] or out.shape[1] < original_shape[1]:
        vd, hd = max(0, original_shape[0] - out.shape[0]), max(0, original_shape[1] - out.shape[1])
        out = np.pad(out, [(vd // 2, vd // 2 + vd % 2), (hd // 2, hd // 2 + hd % 2), (0, 0)], 'constant', constant_values=0.0)
    return out


class SegmentationPrediction(object):
    def __init__(self, final_prediction: torch.Tensor, topk: torch.Tensor):
        self.final_prediction = final_prediction      # [H, W, classes] float32 (device): mean activation of the ensemble
        self.topk = topk                              # [k, H, W] int32 (device): k-th best class per pixel

    def found_classes(self, rank: int = 0):
        """[(class index, fraction of the image)] of the rank-th segmentation, most frequent first
        (the material percentages of src/utils/prediction_utils.py:predictions_to_image)."""
        m = self.topk[rank].reshape(-1).cpu().numpy()
        cnt = np.bincount(m, minlength=self.final_prediction.shape[-1])
        order = np.argsort(-cnt, kind="stable")
        return [(int(c), float(cnt[c]) / m.size) for c in order if cnt[c] > 0]


class SegmentationPredictor(object):
    """The body of src/predict.py:main (465-562) for a loaded model."""

    def __init__(self, model, data_set_information, div2_constraint=4, labeled_only=False):
        self.model = getattr(model, "model", model)      # accepts the ModelBase wrapper or its ExtendedModel
        self.data_set_information = data_set_information
        self.div2_constraint = div2_constraint
        self.labeled_only = labeled_only

    def prediction_images(self, image, ensembling=False, sdim=None) -> List[PredictionImage]:
        if ensembling and (self.model.input_shape[0] is not None and self.model.input_shape[1] is not None):
            raise ValueError('Cannot use dimensional ensembling if the input shape is not variable')
        factors = ENSEMBLING_SCALE_FACTORS if ensembling else [1.0]
        return [PredictionImage(image, self.data_set_information, sdim=sdim, scale_factor=f, div2_constraint=self.div2_constraint,
                                labeled_only=self.labeled_only) for f in factors]

    def predict(self, image, ensembling=False, sdim=None, background_class_prediction_weight=1.0, top_k=1) -> SegmentationPrediction:
        L = lib()
        m = self.model
        dev = m.graph.device
        C = m.num_classes
        pis = self.prediction_images(image, ensembling, sdim)
        H, W = pis[0].original_shape
        acc = torch.empty(H, W, C, dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for i, pi in enumerate(pis):
            logits = m.predict_on_batch_device(pi.device_input(dev))      # [1, ph, pw, C], valid until the next forward
            ph, pw = pi.padded_shape
            h, w = pi.unpadded_shape
            if not pi.scaled:
                if (h, w) != (H, W):
                    raise ValueError('Image shape after undoing transformations does not match the original shape: {} vs {}'.format((h, w), (H, W)))
                L.predict_accumulate(logits.data_ptr(), ph, pw, C, pi.v_pad_before, pi.h_pad_before, h, w,
                                     float(background_class_prediction_weight), int(i > 0), acc.data_ptr(), st)
            else:
                crop = logits[0, pi.v_pad_before:pi.v_pad_before + h, pi.h_pad_before:pi.h_pad_before + w].cpu().numpy()
                back = _zoom_back(crop, pi.scale_factor, (H, W))
                if back.shape[:2] != (H, W):
                    raise ValueError('Image shape after undoing transformations does not match the original shape: {} vs {}'.format(back.shape[:2], (H, W)))
                bt = torch.from_numpy(np.ascontiguousarray(back, dtype=np.float32)).to(dev)
                L.predict_accumulate(bt.data_ptr(), H, W, C, 0, 0, H, W, float(background_class_prediction_weight), int(i > 0),
                                     acc.data_ptr(), st)
                torch.cuda.current_stream(dev).synchronize()
        final = torch.empty_like(acc)
        topk = torch.empty(top_k, H, W, dtype=torch.int32, device=dev)
        L.predict_topk(acc.data_ptr(), H * W, C, len(pis), top_k, final.data_ptr(), topk.data_ptr(), st)
        return SegmentationPrediction(final, topk)
