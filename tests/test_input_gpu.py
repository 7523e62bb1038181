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


def test_pinned_feeder_overlapped_batches_match_the_oracle_and_the_plain_route():
    """Five batches through two slots: every batch lands in the model's input buffers bit for bit as the numpy restatement
    produces it, and training on the fed batches gives the losses of the set_batch route."""
    gen = pkg("generators")
    models, losses = pkg("models"), pkg("losses")
    xo = pkg("keras_extensions.extended_optimizers")
    n, h, w, nc = 2, 32, 32, 24
    rng = np.random.RandomState(5)
    colours = rng.permutation(256)[:nc]
    r_to_class = {int(r): i for i, r in enumerate(colours)}
    cw = rng.uniform(0.5, 2.0, nc).astype(np.float32)
    cw[0] = 0.0
    mean, std = [0.1, -0.05, 0.02], [0.45, 0.5, 0.55]
    batches = [(rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8),
                colours[rng.randint(0, nc, size=(n, h, w))].astype(np.uint8)) for _ in range(5)]

    def make_model():
        m = models.get_model("enet-naive-upsampling", (None, None, 3), nc,
                             losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY, device="cuda:0").model
        m.compile(optimizer=xo.Adam(lr=5e-4))
        m.seed = 99
        return m
    enc = gen.DeviceBatchEncoder(r_to_class, cw, mean, std, True, True)
    nu = np.zeros((n, 1), dtype=np.int32)
    # plain route: encode on the device, stage with set_batch
    ref_model = make_model()
    w0 = ref_model.get_weights()
    ref_losses = []
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for im, mk in batches:
            X, Y, W = enc.encode(im, mk)
            ref_model.set_batch([X, Y, W, nu])
            ref_model.train_step_async()
            stream.synchronize()
            ref_losses.append(float(ref_model._loss_buf.item()))
    # overlapped route
    model = make_model()
    model.set_weights(w0)
    feeder = gen.PinnedBatchFeeder(enc, n, h, w, slots=2)
    got_losses = []
    with pytest.raises(RuntimeError):
        feeder.feed(model)                                  # nothing submitted yet
    with torch.cuda.stream(stream):
        feeder.submit(*batches[0])
        for i in range(len(batches)):
            if i + 1 < len(batches):
                feeder.submit(*batches[i + 1])              # in flight while batch i trains
            feeder.feed(model)
            if i in (0, 3):
                stream.synchronize()
                assert np.array_equal(model.images_buf.cpu().numpy(), O.normalize_image_batch(batches[i][0], mean, std))
                Yr = O.index_encode_batch_masks(batches[i][1], r_to_class, n)
                assert np.array_equal(model.labels_buf.cpu().numpy(), Yr)
                assert np.array_equal(model.weights_buf.cpu().numpy(), O.create_batch_weights(Yr, cw, n))
            model.train_step_async()
            stream.synchronize()
            got_losses.append(float(model._loss_buf.item()))
    enc.check()
    assert got_losses == ref_losses
    feeder.submit(*batches[0])
    feeder.submit(*batches[1])
    with pytest.raises(RuntimeError):
        feeder.submit(*batches[2])                          # both slots hold unfed batches
    with pytest.raises(ValueError):
        gen.PinnedBatchFeeder(enc, n, h, w, slots=2).submit(batches[0][0].astype(np.float32), batches[0][1])
