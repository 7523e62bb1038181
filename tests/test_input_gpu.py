"""Device input pre-pass against the numpy restatement of the reference's generator functions: bit for bit."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import input_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,h,w,nl,mean,std", [
    (3, 17, 23, 3, [0.1, -0.05, 0.02], [0.45, 0.5, 0.55]),
    (4, 32, 32, 2, [120.0, 130.0, 110.0], [200.0, 210.0, 190.0]),     # given in the [0, 255] range
    (2, 8, 8, 2, None, None),
])
def test_preprocess_batch_bit_exact(n, h, w, nl, mean, std):
    gen = pkg("generators")
    rng = np.random.RandomState(n * 7 + h)
    nc = 24
    colours = rng.permutation(256)[:nc]
    r_to_class = {int(r): i for i, r in enumerate(colours)}
    cw = rng.uniform(0.0, 2.0, nc).astype(np.float32)
    cw[0] = 0.0
    images = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    masks = colours[rng.randint(0, nc, size=(n, h, w))].astype(np.uint8)
    enc = gen.DeviceBatchEncoder(r_to_class, cw, mean, std, mean is not None, std is not None)
    X, Y, W = enc.encode(images, masks, nl)
    torch.cuda.synchronize()
    enc.check()
    Xr = O.normalize_image_batch(images, mean, std)
    Yr = O.index_encode_batch_masks(masks, r_to_class, nl)
    Wr = O.create_batch_weights(Yr, cw, nl)
    assert np.array_equal(X.cpu().numpy(), Xr)
    assert np.array_equal(Y.cpu().numpy(), Yr)
    assert np.array_equal(W.cpu().numpy(), Wr)
    # batch assembly of the semi-supervised format
    x, y = enc.get_data_batch(images, masks, nl)
    assert np.array_equal(x[3], np.full(n, n - nl, dtype=np.int32))
    lo = y[1].cpu().numpy()
    assert lo.shape == (n, h, w, 1) and np.array_equal(lo[:nl, ..., 0], Yr[:nl]) and (lo[nl:] == 0).all()


def test_unmapped_colour_is_reported():
    gen = pkg("generators")
    enc = gen.DeviceBatchEncoder({10: 1, 20: 2}, [0.0, 1.0, 1.0], None, None, False, False)
    images = np.zeros((1, 4, 4, 3), dtype=np.uint8)
    masks = np.full((1, 4, 4), 10, dtype=np.uint8)
    masks[0, 1, 2] = 33
    enc.encode(images, masks)
    with pytest.raises(KeyError):
        enc.check()
    with pytest.raises(ValueError):
        gen.DeviceBatchEncoder({10: 1}, [1.0, 1.0], [300.0, 1.0, 1.0], None, True, False)
