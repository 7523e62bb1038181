"""Device side of the reference's batch contract (src/generators.py).

The reference's `SegmentationDataGenerator` crops / augments on CPU workers and then, still on the CPU, normalises the
images (`_np_normalize_image_batch`, :711-770), index-encodes the masks (`_index_encode_batch_masks`, :1787-1812) and
builds the per-pixel weight map (`_create_batch_weights`, :1814-1825).  `DeviceBatchEncoder` keeps those three method
names and semantics but runs them as one HIP pre-pass over the uint8 crops (4 bytes in, 20 bytes out per pixel), so the
host hands over 4 bytes per pixel instead of 20.  File reading, cropping and augmentation stay with the caller.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from ._lib import lib


def np_from_255_to_normalized(val):
    """src/utils/image_utils.py:1437-1441."""
    return ((val / 255.0) - 0.5) * 2.0


def _normalised_range(values, what) -> Optional[np.ndarray]:
    """Per-channel mean / stddev given either in [-1, 1] or in [0, 255] (src/generators.py:741-766)."""
    if values is None:
        return None
    v = np.array(values).astype(np.float32)
    if (v >= -1.0 - 1e-7).all() and (v <= 1.0 + 1e-7).all():
        return v
    if (v >= 0.0).all() and (v <= 255.0).all():
        return np_from_255_to_normalized(v).astype(np.float32)
    raise ValueError('{} is in unknown range: {}'.format(what, v))


class DeviceBatchEncoder(object):
    def __init__(self, material_r_color_to_material_class: Dict[int, int], class_weights: Sequence[float],
                 per_channel_mean=None, per_channel_stddev=None, use_per_channel_mean_normalization=True,
                 use_per_channel_stddev_normalization=True, device="cuda"):
        self.device = torch.device(device)
        if use_per_channel_mean_normalization and per_channel_mean is None:
            raise ValueError('Use per-channel mean normalization is true, but per-channel mean is None')
        if use_per_channel_stddev_normalization and per_channel_stddev is None:
            raise ValueError('Use per-channel stddev normalization is true, but per-channel stddev is None')
        mean = _normalised_range(per_channel_mean, 'Per channel mean') if use_per_channel_mean_normalization else None
        std = _normalised_range(per_channel_stddev, 'Per-channel stddev') if use_per_channel_stddev_normalization else None
        lut = np.full(256, -1, dtype=np.int32)
        for r, cls in material_r_color_to_material_class.items():
            if not 0 <= int(r) <= 255:
                raise ValueError('material red colour {} is not a byte'.format(r))
            lut[int(r)] = int(cls)
        self.num_classes = len(class_weights)
        self._lut = torch.from_numpy(lut).to(self.device)
        self._cw = torch.tensor(np.asarray(class_weights, dtype=np.float32), device=self.device)
        self._mean = torch.from_numpy(mean).to(self.device) if mean is not None else None
        self._std = torch.from_numpy(std).to(self.device) if std is not None else None
        self._bad = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _u8(self, a) -> torch.Tensor:
        if torch.is_tensor(a):
            if a.dtype != torch.uint8:
                raise ValueError('expected uint8 data, got {}'.format(a.dtype))
            return a.to(self.device).contiguous()
        a = np.asarray(a)
        if a.dtype != np.uint8:
            if a.min() < 0 or a.max() > 255:
                raise ValueError('Batch image values are not between [0, 255], got [{}, {}]'.format(a.min(), a.max()))
            a = a.astype(np.uint8)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def encode(self, images, masks=None, num_labeled: Optional[int] = None, stream: Optional[int] = None):
        """images uint8 [n,h,w,3]; masks uint8 [n,h,w] (red channel) or None.
        Returns (X float32 [n,h,w,3], Y int32 [n,h,w] | None, W float32 [n,h,w] | None) as device tensors."""
        img = self._u8(images)
        if img.dim() != 4 or img.shape[3] != 3:
            raise ValueError('images must be [n, h, w, 3], got {}'.format(tuple(img.shape)))
        n, h, w = int(img.shape[0]), int(img.shape[1]), int(img.shape[2])
        X = torch.empty(n, h, w, 3, dtype=torch.float32, device=self.device)
        Y = W = msk = None
        if masks is not None:
            msk = self._u8(masks)
            if tuple(msk.shape) != (n, h, w):
                raise ValueError('masks must be [n, h, w] = {}, got {}'.format((n, h, w), tuple(msk.shape)))
            Y = torch.empty(n, h, w, dtype=torch.int32, device=self.device)
            W = torch.empty(n, h, w, dtype=torch.float32, device=self.device)
        nl = n if num_labeled is None else int(num_labeled)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        p = lambda t: None if t is None else t.data_ptr()
        lib().preprocess_batch(p(img), p(msk), p(self._lut), p(self._cw), self.num_classes, nl, p(self._mean), p(self._std),
                               n, h, w, p(X), p(Y), p(W), p(self._bad), st)
        return X, Y, W

    def check(self):
        """Raises KeyError (like the reference's dictionary lookup) if a labeled pixel had a colour without a class."""
        bad = int(self._bad.item())
        if bad:
            self._bad.zero_()
            raise KeyError('{} labeled pixels carry a material colour that maps to no class'.format(bad))

    def get_data_batch(self, images, masks, num_labeled: Optional[int] = None):
        """The SEMI_SUPERVISED batch format of src/generators.py:1742-1766:
        ([X, Y, W, num_unlabeled], [dummy_output, logits_output])."""
        X, Y, W = self.encode(images, masks, num_labeled)
        n = int(X.shape[0])
        nl = n if num_labeled is None else int(num_labeled)
        num_unlabeled = np.ones(n, dtype=np.int32) * (n - nl)
        logits_output = Y.clone().unsqueeze(-1)
        logits_output[nl:] = 0
        return [X, Y, W, num_unlabeled], [np.zeros(n, dtype=np.int32), logits_output]


class PinnedBatchFeeder(object):
    """Overlaps the host -> device transfer of the NEXT batch with the training step of the current one.

    A generator worker of the reference produces uint8 crops and uint8 masks and the Python side turns them into the
    float32 batch of src/generators.py:1742-1766 on the host, 84 MB per batch of 16 at 512x512.  Here the uint8 arrays
    (25 MB) are copied into one of `slots` pinned host buffers, sent on a copy stream, and `feed` runs the fused
    preprocess kernel from the device copy straight into the static input buffers of the model's captured step, after
    the previous step on the same stream.  Only the supervised batch format [X, Y, W, num_unlabeled = 0] is fed."""

    class _Slot(object):
        pass

    def __init__(self, encoder: DeviceBatchEncoder, n: int, h: int, w: int, slots: int = 2):
        if slots < 2:
            raise ValueError('at least two slots are needed to overlap a transfer with a step')
        self.enc, self.shape = encoder, (int(n), int(h), int(w))
        dev = encoder.device
        self.copy_stream = torch.cuda.Stream(device=dev)
        self._slots = []
        for _ in range(slots):
            s = PinnedBatchFeeder._Slot()
            s.host_img = torch.empty((n, h, w, 3), dtype=torch.uint8).pin_memory()
            s.host_msk = torch.empty((n, h, w), dtype=torch.uint8).pin_memory()
            s.dev_img = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
            s.dev_msk = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
            s.sent = torch.cuda.Event()        # the transfer into the device copy has finished
            s.consumed = torch.cuda.Event()    # the preprocess kernel has read the device copy
            s.sent_once = s.consumed_once = False
            self._slots.append(s)
        self._submitted = self._fed = 0

    def pending(self) -> int:
        return self._submitted - self._fed

    def submit(self, images, masks):
        """images uint8 [n,h,w,3], masks uint8 [n,h,w] (numpy).  Returns once the arrays are in pinned memory; the
        transfer itself is asynchronous."""
        if self.pending() >= len(self._slots):
            raise RuntimeError('every slot holds a batch that has not been fed yet')
        n, h, w = self.shape
        images, masks = np.asarray(images), np.asarray(masks)
        if images.dtype != np.uint8 or masks.dtype != np.uint8:
            raise ValueError('expected uint8 images and masks, got {} and {}'.format(images.dtype, masks.dtype))
        if images.shape != (n, h, w, 3) or masks.shape != (n, h, w):
            raise ValueError('expected images {} and masks {}, got {} and {}'.format((n, h, w, 3), (n, h, w), images.shape,
                                                                                    masks.shape))
        s = self._slots[self._submitted % len(self._slots)]
        if s.sent_once:
            s.sent.synchronize()               # the pinned buffers are free again once their last transfer is done
        np.copyto(s.host_img.numpy(), images)
        np.copyto(s.host_msk.numpy(), masks)
        with torch.cuda.stream(self.copy_stream):
            if s.consumed_once:
                self.copy_stream.wait_event(s.consumed)
            s.dev_img.copy_(s.host_img, non_blocking=True)
            s.dev_msk.copy_(s.host_msk, non_blocking=True)
            s.sent.record(self.copy_stream)
        s.sent_once = True
        self._submitted += 1

    def feed(self, model):
        """Queues, on the current stream, the preprocess of the oldest submitted batch into `model`'s input buffers."""
        if self.pending() <= 0:
            raise RuntimeError('no submitted batch to feed')
        if model.mean_teacher or model.superpixel or model.classification or not model.lambda_loss:
            raise ValueError('PinnedBatchFeeder feeds the supervised segmentation batch format only')
        n, h, w = self.shape
        model._ensure_buffers(n, h, w)
        s = self._slots[self._fed % len(self._slots)]
        main = torch.cuda.current_stream(self.enc.device)
        main.wait_event(s.sent)
        e = self.enc
        p = lambda t: None if t is None else t.data_ptr()
        lib().preprocess_batch(p(s.dev_img), p(s.dev_msk), p(e._lut), p(e._cw), e.num_classes, n, p(e._mean), p(e._std),
                               n, h, w, p(model.images_buf), p(model.labels_buf), p(model.weights_buf), p(e._bad),
                               main.cuda_stream)
        s.consumed.record(main)
        s.consumed_once = True
        self._fed += 1


def calculate_mfb_class_weights(class_probabilities):
    """Median-frequency-balancing class weights (src/utils/dataset_utils.py:498-513): median of the non-zero class
    probabilities divided by each probability, 0 for classes that never occur."""
    p = np.array(class_probabilities, dtype=np.float64)
    med = np.median(p[p > 0.0])
    return [float(med / v) if v > 0.0 else 0.0 for v in p]


def calculate_enet_class_weights(class_probabilities, c=1.02):
    """ENet class weights (src/utils/dataset_utils.py:515-529): 1 / ln(c + p), 0 for classes that never occur."""
    return [float(1.0 / np.log(c + v)) if v > 0.0 else 0.0 for v in np.array(class_probabilities)]
