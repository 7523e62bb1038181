"""Full-image inference (semantic-segmentation_amd/predict.py) against the numpy oracle of src/predict.py.

Bars: the network input (normalise + pad) and everything after the network (crop, background weight, ensemble mean,
top-k) are bit-exact against the oracle applied to the same logits; the end-to-end mean prediction is within rtol 1e-3 of
the fp64 oracle network run on the oracle's input."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E
from oracle import predict_oracle as P

pytestmark = pytest.mark.gpu
MEAN = [0.12, -0.05, 0.2]
STD = [0.55, 0.6, 0.5]


def _model(name="enet-naive-upsampling", nc=24, seed=4):
    models = pkg("models")
    mb = models.get_model(name, (None, None, 3), nc)
    w0 = E.init_weights(name, nc, seed=seed, perturb=True)
    mb.model.set_weights([w0[k] for k in mb.model.weight_names()])
    return mb, w0


@pytest.mark.parametrize("shape,div2", [((37, 50), 4), ((48, 64), 4), ((33, 31), 5), ((7, 9), 4)])
def test_prepare_is_bit_exact(shape, div2):
    pr = pkg("predict")
    rng = np.random.RandomState(shape[0])
    img = rng.randint(0, 256, size=shape + (3,)).astype(np.uint8)
    dsi = pr.DataSetInformation(MEAN, STD)
    pi = pr.PredictionImage(img, dsi, div2_constraint=div2)
    X = pi.device_input("cuda")
    ref, pads = P.prediction_input(img, MEAN, STD, np.array(MEAN, np.float32), div2)
    assert (pi.v_pad_before, pi.v_pad_after, pi.h_pad_before, pi.h_pad_after) == pads
    assert tuple(X.shape[1:3]) == ref.shape[:2] and pi.padded == (ref.shape[:2] != shape)
    assert np.array_equal(X[0].cpu().numpy(), ref)
    # statistics given in the [0, 255] range are converted first (src/utils/image_utils.py:1486-1490)
    pi2 = pr.PredictionImage(img.astype(np.float32), pr.DataSetInformation([120.0, 130.0, 140.0], None), div2_constraint=div2)
    m = (((np.array([120.0, 130.0, 140.0], np.float32) / 255.0) - 0.5) * 2.0).astype(np.float32)
    ref2, _ = P.prediction_input(img, m, None, np.array([120.0, 130.0, 140.0], np.float32), div2)
    assert np.array_equal(pi2.device_input("cuda")[0].cpu().numpy(), ref2)


def test_topk_ties_rank_lower_class_first():
    L = pkg("_lib").lib()
    acc = torch.tensor([[1.0, 3.0, 3.0, -2.0, 3.0], [0.0, 0.0, 0.0, 0.0, 0.0]], device="cuda")
    topk = torch.empty(5, 2, dtype=torch.int32, device="cuda")
    mean = torch.empty_like(acc)
    L.predict_topk(acc.data_ptr(), 2, 5, 3, 5, mean.data_ptr(), topk.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert topk[:, 0].tolist() == [1, 2, 4, 0, 3] and topk[:, 1].tolist() == [0, 1, 2, 3, 4]
    assert np.array_equal(mean.cpu().numpy(), acc.cpu().numpy() / np.float32(3))


@pytest.mark.parametrize("shape,bgw,k", [((37, 50), 0.6, 3), ((32, 48), 1.0, 1)])
def test_single_scale_pipeline(shape, bgw, k):
    pr = pkg("predict")
    mb, w0 = _model()
    rng = np.random.RandomState(11)
    img = rng.randint(0, 256, size=shape + (3,)).astype(np.uint8)
    dsi = pr.DataSetInformation(MEAN, STD)
    res = pr.SegmentationPredictor(mb, dsi, div2_constraint=4).predict(img, background_class_prediction_weight=bgw, top_k=k)
    torch.cuda.synchronize()
    x, pads = P.prediction_input(img, MEAN, STD, np.array(MEAN, np.float32), 4)
    # (1) the part after the network, on the logits the kernels produced: bit-exact
    hip_logits = mb.model.graph.logits[0].cpu().numpy()
    final, topk = P.finish([hip_logits], [pads], bgw, k)
    assert np.array_equal(res.final_prediction.cpu().numpy(), final)
    assert np.array_equal(res.topk.cpu().numpy(), topk)
    assert res.final_prediction.shape == shape + (24,)
    # (2) end to end against the fp64 oracle network in its inference phase
    params = {k_: torch.tensor(np.asarray(v), dtype=torch.float64) for k_, v in w0.items()}
    with torch.no_grad():
        ref_logits = E.enet_forward(mb_name(mb), params, torch.tensor(x[None], dtype=torch.float64), training=False)[0][0].numpy()
    ref_final, ref_topk = P.finish([ref_logits], [pads], bgw, k)
    err = np.abs(res.final_prediction.cpu().numpy() - ref_final)
    assert (err <= 1e-3 * np.abs(ref_final) + 1e-4 * np.abs(ref_final).max()).all()
    srt = np.sort(ref_final, axis=-1)
    near_ties = int(((srt[..., -1] - srt[..., -2]) < 1e-4 * np.abs(ref_final).max()).sum())
    assert int((res.topk[0].cpu().numpy() != ref_topk[0]).sum()) <= near_ties
    fc = res.found_classes(0)
    assert abs(sum(f for _, f in fc) - 1.0) < 1e-9 and fc[0][1] >= fc[-1][1]


def mb_name(mb):
    return {"ENet-Naive-Upsampling": "enet-naive-upsampling", "ENet-Max-Unpooling": "enet-max-unpooling"}[mb.name]


def test_ensembling_pipeline_matches_numpy_composition():
    """Three scales (src/predict.py:454): per scale the model's own logits are cropped, zoomed back with scipy, weighted and
    averaged with numpy exactly as src/predict.py:482-527 does; the device result must match bit for bit."""
    from scipy import ndimage
    pr = pkg("predict")
    mb, _ = _model("enet-max-unpooling", seed=6)
    rng = np.random.RandomState(2)
    img = np.clip(rng.randint(0, 256, size=(40, 56, 3)) // 32 * 32 + 8, 0, 255).astype(np.uint8)
    dsi = pr.DataSetInformation(MEAN, STD)
    sp = pr.SegmentationPredictor(mb, dsi, div2_constraint=4)
    res = sp.predict(img, ensembling=True, background_class_prediction_weight=0.8, top_k=2)
    torch.cuda.synchronize()
    preds = []
    for pi in sp.prediction_images(img, ensembling=True):
        lg = mb.model.predict_on_batch(pi.device_input("cuda"))[0]
        h, w = pi.unpadded_shape
        p = lg[pi.v_pad_before:pi.v_pad_before + h, pi.h_pad_before:pi.h_pad_before + w]
        if pi.scaled:
            sf = 1.0 / pi.scale_factor
            p = ndimage.zoom(p, zoom=(sf, sf, 1.0), order=3, mode='constant', cval=-1.0).astype(p.dtype)
            H, W = pi.original_shape
            if p.shape[0] > H:
                p = p[p.shape[0] - H:, :, :]
            if p.shape[1] > W:
                p = p[:, p.shape[1] - W:, :]
            assert p.shape[:2] == (H, W)
        p = np.array(p, dtype=np.float32)
        p[:, :, 0] *= 0.8
        preds.append(p)
    final = np.mean(np.array(preds), axis=0)
    assert final.dtype == np.float32
    assert np.array_equal(res.final_prediction.cpu().numpy(), final)
    order = np.argsort(-final, kind="stable")
    assert np.array_equal(res.topk.cpu().numpy(), np.transpose(np.transpose(order)[:2], axes=(0, 2, 1)))
