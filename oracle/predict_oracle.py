"""CPU oracle of the full-image inference path: numpy restatement of src/predict.py:86-125 (PredictionImage), :164-188
(pad_image_to_div2_constraint), :465-527 (predict, crop, background weight, mean) and
src/utils/prediction_utils.py:95-99 (top-k ranking).

TEST INFRASTRUCTURE ONLY (tests/).  PARITY UNPINNED: the reference holds no test vectors for these functions; they are
plain numpy, restated operation by operation in float32, so everything around the network is compared bit for bit.
"""
import numpy as np


def normalize_image_channels(img, per_channel_mean=None, per_channel_stddev=None):
    """src/utils/image_utils.py:1451-1520 for statistics already in the [-1, 1] range."""
    x = np.array(img).astype(np.float32)
    if np.min(x) < 0 or np.max(x) > 255:
        raise ValueError('Image values are not in range [0, 255]')
    x = ((x / 255.0) - 0.5) * 2.0
    if per_channel_mean is not None:
        x -= np.array(per_channel_mean).astype(np.float32)
    if per_channel_stddev is not None:
        x /= np.array(per_channel_stddev).astype(np.float32)
    return x


def trailing_zeroes(num):
    z = 0
    while ((num >> z) & 1) == 0:
        z += 1
    return z


def closest_higher_with_trailing_zeroes(num, n):
    """src/utils/dataset_utils.py:1338-1361."""
    if trailing_zeroes(num) >= n:
        return num
    unit = 1 << n
    if num < unit:
        return unit
    return num + (unit - (num % unit))


def pad_image_to_div2_constraint(image_array, div2_constraint, cval):
    """src/predict.py:164-188 + src/utils/image_utils.py:1400-1434 (pad with a sentinel, then paint it with cval)."""
    ph = closest_higher_with_trailing_zeroes(image_array.shape[0], div2_constraint)
    pw = closest_higher_with_trailing_zeroes(image_array.shape[1], div2_constraint)
    vd, hd = max(0, ph - image_array.shape[0]), max(0, pw - image_array.shape[1])
    vb, va = vd // 2, (vd // 2) + (vd % 2)
    hb, ha = hd // 2, (hd // 2) + (hd % 2)
    tmp = 919191.0
    out = np.pad(image_array, [(vb, va), (hb, ha), (0, 0)], 'constant', constant_values=tmp)
    out[out[:, :, 0] == tmp] = cval
    return out, vb, va, hb, ha


def prediction_input(img, mean, std, cval, div2_constraint):
    """(network input HxWx3 float32, pads) of src/predict.py:115-125."""
    x = normalize_image_channels(img, mean, std)
    pads = (0, 0, 0, 0)
    if trailing_zeroes(x.shape[0]) < div2_constraint or trailing_zeroes(x.shape[1]) < div2_constraint:
        x, vb, va, hb, ha = pad_image_to_div2_constraint(x, div2_constraint, cval)
        pads = (vb, va, hb, ha)
    return x, pads


def finish(predictions, pads_list, bg_weight, k):
    """src/predict.py:482-527 without rescaling + prediction_utils.py:95-99: (mean prediction, top-k class maps [k,H,W])."""
    outs = []
    for p, (vb, va, hb, ha) in zip(predictions, pads_list):
        p = np.array(p, dtype=np.float32)
        p = p[vb:p.shape[0] - va, hb:p.shape[1] - ha]
        p[:, :, 0] *= bg_weight
        outs.append(p)
    final = np.mean(np.array(outs), axis=0)
    order = np.argsort(-final, kind="stable")
    topk = np.transpose(np.transpose(order)[:k], axes=(0, 2, 1))
    return final, topk
