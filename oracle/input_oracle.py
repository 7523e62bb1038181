"""CPU oracle of the input contract: numpy restatement of src/generators.py:711-770 (_np_normalize_image_batch),
:1787-1812 (_index_encode_batch_masks) and :1814-1825 (_create_batch_weights).

TEST INFRASTRUCTURE ONLY (tests/).  PARITY UNPINNED: the reference holds no test vectors for these functions; they are
plain numpy, restated here operation by operation (same float32 in-place order), so the device result is compared
bit for bit.
"""
import numpy as np


def normalize_image_batch(batch, per_channel_mean=None, per_channel_stddev=None):
    batch = batch.astype(np.float32)
    if np.min(batch) < 0 or np.max(batch) > 255:
        raise ValueError('Batch image values are not between [0, 255]')
    batch /= 255.0
    batch -= 0.5
    batch *= 2.0
    if per_channel_mean is not None:
        m = np.array(per_channel_mean).astype(np.float32)
        if (m >= -1.0 - 1e-7).all() and (m <= 1.0 + 1e-7).all():
            batch -= m
        else:
            batch -= (((m / 255.0) - 0.5) * 2.0).astype(np.float32)
    if per_channel_stddev is not None:
        s = np.array(per_channel_stddev).astype(np.float32)
        if (s >= -1.0 - 1e-7).all() and (s <= 1.0 + 1e-7).all():
            batch /= s
        else:
            batch /= (((s / 255.0) - 0.5) * 2.0).astype(np.float32)
    return batch


def index_encode_batch_masks(masks_r, r_to_class, num_labeled):
    out = np.zeros(masks_r.shape, dtype=np.int32)
    for i in range(masks_r.shape[0]):
        m = masks_r[i].astype(np.int32)
        if i < num_labeled:
            for r in np.unique(m):
                out[i][m == r] = r_to_class[int(r)]
        else:
            out[i] = m
    return out


def create_batch_weights(np_masks, class_weights, num_labeled):
    w = np.ones_like(np_masks, dtype=np.float32)
    lw, lm = w[0:num_labeled], np_masks[0:num_labeled]
    for c in np.unique(lm):
        lw[lm == c] = class_weights[c]
    return w
